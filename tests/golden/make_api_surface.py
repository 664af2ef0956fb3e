"""Generates tests/golden/api_surface.json: the parameter lists of the reference's public callables on the learner path, read
from the reference's SOURCE with `ast` (nothing is imported or copied; names and parameter order only).  Run in the build
container, where /root/reference exists:  python tests/golden/make_api_surface.py"""
import ast
import json
import os

REF = "/root/reference/serl_launcher/serl_launcher"
FILES = ["agents/continuous/drq.py", "agents/continuous/sac.py", "data/data_store.py", "utils/launcher.py",
         "utils/train_utils.py", "networks/reward_classifier.py", "data/memory_efficient_replay_buffer.py",
         "data/replay_buffer.py", "data/dataset.py"]


def params(f):
    a = f.args
    pos = a.posonlyargs + a.args
    ndef = len(a.defaults)
    out = []
    for i, x in enumerate(pos):
        out.append({"name": x.arg, "kind": "positional", "default": i >= len(pos) - ndef})
    if a.vararg:
        out.append({"name": a.vararg.arg, "kind": "var_positional", "default": False})
    for x, d in zip(a.kwonlyargs, a.kw_defaults):
        out.append({"name": x.arg, "kind": "keyword_only", "default": d is not None})
    if a.kwarg:
        out.append({"name": a.kwarg.arg, "kind": "var_keyword", "default": False})
    return out


def main():
    surface = {}
    for rel in FILES:
        path = os.path.join(REF, rel)
        tree = ast.parse(open(path).read())
        entries = {}
        for node in tree.body:
            if isinstance(node, ast.FunctionDef) and not node.name.startswith("_"):
                entries[node.name] = {"line": node.lineno, "params": params(node)}
            if isinstance(node, ast.ClassDef):
                for m in node.body:
                    if isinstance(m, ast.FunctionDef) and (not m.name.startswith("_") or m.name == "__init__"):
                        entries[f"{node.name}.{m.name}"] = {"line": m.lineno, "params": params(m)}
        surface[rel] = entries
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "api_surface.json")
    json.dump(surface, open(out, "w"), indent=1, sort_keys=True)
    print("wrote", out, sum(len(v) for v in surface.values()), "callables")


if __name__ == "__main__":
    main()
