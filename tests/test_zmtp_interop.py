"""CPU: serl_amd/transport/zmtp.py against a REAL libzmq (the library under pyzmq / agentlace), both directions, for the two
patterns the actor uses: REQ/REP and PUB/SUB.  Runs wherever a libzmq shared object is loadable (SERL_LIBZMQ=<path>,
or the system's libzmq via ctypes.util.find_library); skipped otherwise -- this image ships none, the recorded run is
profiles/r03_zmtp_interop.txt."""
import ctypes as C
import ctypes.util
import os
import socket
import threading
import time

import pytest

from serl_amd.transport import zmtp as Z


def _load():
    for name in (os.environ.get("SERL_LIBZMQ"), ctypes.util.find_library("zmq")):
        if not name:
            continue
        try:
            L = C.CDLL(name)
        except OSError:
            continue
        L.zmq_ctx_new.restype = C.c_void_p
        L.zmq_socket.restype = C.c_void_p
        L.zmq_socket.argtypes = [C.c_void_p, C.c_int]
        L.zmq_bind.argtypes = L.zmq_connect.argtypes = [C.c_void_p, C.c_char_p]
        L.zmq_send.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int]
        L.zmq_recv.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        L.zmq_setsockopt.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        L.zmq_close.argtypes = [C.c_void_p]
        L.zmq_version.argtypes = [C.POINTER(C.c_int)] * 3
        return L
    return None


L = _load()
pytestmark = pytest.mark.skipif(L is None, reason="no libzmq shared object on this host (set SERL_LIBZMQ to test against one)")
SIZES = (5, 300, 70000, 3_000_000)


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _zsock(ctx, typ):
    s = L.zmq_socket(ctx, typ)
    for opt in (27, 28):     # ZMQ_RCVTIMEO, ZMQ_SNDTIMEO
        L.zmq_setsockopt(s, opt, C.byref(C.c_int(10000)), 4)
    L.zmq_setsockopt(s, 17, C.byref(C.c_int(0)), 4)   # ZMQ_LINGER
    return s


def _zrecv(s, cap=4 << 20):
    b = C.create_string_buffer(cap)
    n = L.zmq_recv(s, b, cap, 0)
    assert n >= 0, "libzmq recv failed / timed out"
    return b.raw[:n]


def test_version():
    v = [C.c_int() for _ in range(3)]
    L.zmq_version(*[C.byref(x) for x in v])
    print("libzmq", ".".join(str(x.value) for x in v))
    assert v[0].value >= 4


def test_libzmq_req_to_zmtp_rep():
    ctx = L.zmq_ctx_new()
    rep = Z.Context.instance().socket(Z.REP)
    rep.bind("tcp://127.0.0.1:0")

    def serve():
        for _ in SIZES:
            rep.send(b"E:" + rep.recv())
    th = threading.Thread(target=serve)
    th.start()
    rq = _zsock(ctx, 3)
    assert L.zmq_connect(rq, f"tcp://127.0.0.1:{rep.port}".encode()) == 0
    for n in SIZES:
        m = bytes([n % 251]) * n
        assert L.zmq_send(rq, m, n, 0) == n
        assert _zrecv(rq) == b"E:" + m
    th.join()
    L.zmq_close(rq)
    rep.close()


def test_zmtp_req_to_libzmq_rep():
    ctx = L.zmq_ctx_new()
    port = _port()
    rp = _zsock(ctx, 4)
    assert L.zmq_bind(rp, f"tcp://127.0.0.1:{port}".encode()) == 0

    def serve():
        for _ in SIZES:
            out = b"R:" + _zrecv(rp)
            L.zmq_send(rp, out, len(out), 0)
    th = threading.Thread(target=serve)
    th.start()
    rq = Z.Context.instance().socket(Z.REQ)
    rq.setsockopt(Z.RCVTIMEO, 10000)
    rq.connect(f"tcp://127.0.0.1:{port}")
    for n in SIZES:
        m = bytes([n % 249]) * n
        rq.send(m)
        assert rq.recv() == b"R:" + m
    th.join()
    rq.close()
    L.zmq_close(rp)


def test_zmtp_pub_to_libzmq_sub():
    ctx = L.zmq_ctx_new()
    pb = Z.Context.instance().socket(Z.PUB)
    pb.bind("tcp://127.0.0.1:0")
    sb = _zsock(ctx, 2)
    L.zmq_setsockopt(sb, 6, b"", 0)    # ZMQ_SUBSCRIBE ""
    assert L.zmq_connect(sb, f"tcp://127.0.0.1:{pb.port}".encode()) == 0
    got = []
    th = threading.Thread(target=lambda: got.append(_zrecv(sb)))
    th.start()
    for i in range(400):
        pb.send(b"params-%03d" % i + b"p" * 100000)
        if got:
            break
        time.sleep(0.02)
    th.join()
    assert got and got[0].startswith(b"params-") and len(got[0]) == 100010
    L.zmq_close(sb)
    pb.close()


def test_libzmq_pub_to_zmtp_sub():
    ctx = L.zmq_ctx_new()
    port = _port()
    pb = _zsock(ctx, 1)
    assert L.zmq_bind(pb, f"tcp://127.0.0.1:{port}".encode()) == 0
    sb = Z.Context.instance().socket(Z.SUB)
    sb.setsockopt(Z.SUBSCRIBE, b"")
    sb.connect(f"tcp://127.0.0.1:{port}")
    for i in range(400):
        m = b"w-%03d" % i + b"w" * 100000
        L.zmq_send(pb, m, len(m), 0)
        if sb.poll(20):
            break
    m = sb.recv()
    assert m.startswith(b"w-") and len(m) == 100005
    sb.close()
    L.zmq_close(pb)
