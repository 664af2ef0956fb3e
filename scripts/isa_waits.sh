#!/bin/bash
# ISA audit: global loads / stores vs `s_waitcnt vmcnt(0)` per kernel of one .hip file (no GPU needed).
# usage: scripts/isa_waits.sh serl_amd/csrc/heads.hip   -- a kernel whose vmcnt(0) count approaches its load count waits for every load at once
R=$(cd "$(dirname "$0")/.." && pwd); T=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -S --cuda-device-only -I$R/include -I$R/serl_amd/csrc "$1" -o $T/k.s 2>/dev/null
python3 - $T/k.s <<'PY'
import re, sys
cur, stats = None, {}
for l in open(sys.argv[1]):
    m = re.match(r'^(_ZN4serl\S+):\s', l)
    if m:
        cur = m.group(1); stats[cur] = dict(n=0, ld=0, st=0, w0=0, wn=0); continue
    if cur is None: continue
    if '.amdhsa_kernel' in l or l.startswith('.Lfunc_end'): cur = None; continue
    t = l.strip()
    if not t or t[0] in ';.': continue
    s = stats[cur]; s['n'] += 1
    if re.match(r'(global|buffer|flat)_load', t): s['ld'] += 1
    elif re.match(r'(global|buffer|flat)_(store|atomic)', t): s['st'] += 1
    elif t.startswith('s_waitcnt') and 'vmcnt(0)' in t: s['w0'] += 1
    elif t.startswith('s_waitcnt') and 'vmcnt' in t: s['wn'] += 1
for k, v in stats.items():
    print(f"{re.sub(r'^_ZN4serl[0-9]+', '', k)[:72]:74s} instr={v['n']:5d} loads={v['ld']:3d} stores={v['st']:3d} vmcnt(0)={v['w0']:3d} vmcnt(n)={v['wn']:3d}")
PY
rm -rf $T
