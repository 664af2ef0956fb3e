"""CPU: the actor <-> learner endpoint over REAL TCP sockets (transport="zmq"): REQ/REP datastore shipping + requests and
the PUB/SUB network broadcast of examples/async_drq_sim/async_drq_sim.py:95-108,161-171,202-229,297, i.e. the code path
that runs between an actor machine and the learner (reference wiring: utils/launcher.py:171-177, data_store.py:83-144).
The socket layer is pyzmq when installed, else serl_amd/transport/zmtp.py (ZeroMQ's wire protocol in pure Python)."""
import pickle
import socket
import threading
import time

import numpy as np

from serl_amd.transport import DataStoreBase, QueuedDataStore, TrainerClient, TrainerServer, make_trainer_config
from serl_amd.transport import endpoint as E


class ListStore(DataStoreBase):
    def __init__(self, capacity):
        super().__init__(capacity)
        self.items, self.threads = [], set()

    def insert(self, data):
        self.items.append(data)
        self.threads.add(threading.current_thread().name)

    def latest_data_id(self):
        return len(self.items)

    def get_latest_data(self, from_id):
        raise NotImplementedError

    def __len__(self):
        return len(self.items)


def _tr(k):
    return {"observations": {"state": np.full((1, 3), k, np.float32), "front": np.full((1, 128, 128, 3), k % 256, np.uint8)},
            "actions": np.zeros(2, np.float32), "rewards": np.float32(k), "masks": np.float32(1), "dones": False}


def _free_ports():
    out = []
    for _ in range(2):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        out.append((s, s.getsockname()[1]))
    for s, _ in out:
        s.close()
    return out[0][1], out[1][1]


def test_frames_are_real_lz4():
    assert E.LZ4_BACKEND != "zlib (no lz4 on this host)", "neither the lz4 package nor liblz4 is loadable here"
    frame = E.encode({"type": "datastore", "payload": [_tr(1)]})
    assert frame[:4] == b"\x04\x22\x4d\x18"            # LZ4 frame magic number (little-endian 0x184D2204)
    assert len(frame) < 5000                             # 49 KB of constant pixels compress
    out = E.decode(frame)
    assert np.array_equal(out["payload"][0]["observations"]["front"], _tr(1)["observations"]["front"])


def test_actor_learner_flow_over_tcp():
    port, bport = _free_ports()
    cfg = make_trainer_config(port_number=port, broadcast_port=bport)
    stats, nets = [], []
    store = ListStore(1000)
    server = TrainerServer(cfg, request_callback=lambda t, p: stats.append((t, p)) or {"ack": len(stats)}, transport="zmq",
                           bind_ip="127.0.0.1")
    server.register_data_store("actor_env", store)
    server.start(threaded=True)
    try:
        local = QueuedDataStore(2000)
        client = TrainerClient("actor_env", "127.0.0.1", cfg, local, wait_for_server=True, transport="zmq", timeout=10.0)
        client.recv_network_callback(lambda p: nets.append(p))
        for k in range(25):
            local.insert(_tr(k))
            if k % 10 == 9:
                assert client.update()
        assert len(store) == 20 and [float(d["rewards"]) for d in store.items] == list(range(20))
        assert store.threads == {"TrainerServer"}       # inserted by the server thread, like agentlace does
        assert np.array_equal(store.items[7]["observations"]["front"], _tr(7)["observations"]["front"])
        assert client.request("send-stats", {"episode_return": 1.5}) == {"ack": 1}
        assert stats == [("send-stats", {"episode_return": 1.5})]
        assert client.request("no-such-type", {}) is None
        params = {"modules_actor": {"Dense_0": {"kernel": np.arange(12, dtype=np.float32).reshape(3, 4)}}}
        deadline = time.time() + 10
        while not nets and time.time() < deadline:      # PUB/SUB: a subscriber only gets what is published after it joined
            server.publish_network(params)
            time.sleep(0.05)
        assert nets and np.array_equal(nets[0]["modules_actor"]["Dense_0"]["kernel"], params["modules_actor"]["Dense_0"]["kernel"])
        client.stop()
    finally:
        server.stop()
    assert server.stats["transitions"] == 20 and server.stats["requests"] == 1


def test_malformed_frame_gets_an_error_reply_and_the_server_survives():
    """an exception in decode / handle must not kill the REP loop (every actor would block forever)."""
    port, bport = _free_ports()
    cfg = make_trainer_config(port_number=port, broadcast_port=bport)
    store = ListStore(10)

    class Bad(DataStoreBase):
        def __init__(self):
            super().__init__(1)

        def insert(self, data):
            raise RuntimeError("store exploded")

    server = TrainerServer(cfg, transport="zmq", bind_ip="127.0.0.1")
    server.register_data_store("ok", store)
    server.register_data_store("bad", Bad())
    server.start(threaded=True)
    try:
        req = E._zmq.Context.instance().socket(E._zmq.REQ)
        req.setsockopt(E._zmq.RCVTIMEO, 5000)
        req.connect(f"tcp://127.0.0.1:{port}")
        req.send(b"this is not an lz4 frame")
        r = E.decode(req.recv())
        assert r["success"] is False and r["message"]
        req.send(E.encode({"type": "datastore", "store_name": "bad", "payload": [1]}))
        r = E.decode(req.recv())
        assert r["success"] is False and "store exploded" in r["message"]
        req.send(E.encode({"type": "datastore", "store_name": "ok", "payload": [_tr(0)]}))   # still alive
        assert E.decode(req.recv())["success"] is True and len(store) == 1
        req.close(0)
        assert server.stats["errors"] == 2
    finally:
        server.stop()


def test_update_ships_every_transition_exactly_once_under_concurrent_inserts():
    port, bport = _free_ports()
    cfg = make_trainer_config(port_number=port, broadcast_port=bport)
    store = ListStore(100000)
    server = TrainerServer(cfg, transport="zmq", bind_ip="127.0.0.1")
    server.register_data_store("actor_env", store)
    server.start(threaded=True)
    try:
        local = QueuedDataStore(100000)
        client = TrainerClient("actor_env", "127.0.0.1", cfg, local, wait_for_server=True, transport="zmq", timeout=10.0)
        stop = threading.Event()

        def env_loop():
            k = 0
            while not stop.is_set() and k < 3000:
                local.insert({"rewards": np.float32(k)})
                k += 1
        th = threading.Thread(target=env_loop)
        th.start()
        while th.is_alive():
            assert client.update()
        th.join()
        assert client.update()
        got = [int(d["rewards"]) for d in store.items]
        assert got == list(range(len(got))) and len(got) == local.latest_data_id() + 1
        client.stop()
    finally:
        stop.set()
        server.stop()


def test_client_without_a_server_times_out_cleanly():
    port, bport = _free_ports()
    cfg = make_trainer_config(port_number=port, broadcast_port=bport)
    t0 = time.time()
    try:
        TrainerClient("actor_env", "127.0.0.1", cfg, QueuedDataStore(10), transport="zmq", timeout=0.5)
    except ConnectionError:
        pass
    else:
        raise AssertionError("handshake without a server must fail")
    assert time.time() - t0 < 10


def test_pub_never_blocks_on_a_stalled_subscriber_and_never_tears_a_frame():
    """a subscriber that stops reading must cost the publisher whole messages (or the link), never 30 s per
    send and never a partial frame followed by the next message (zmtp.py PUB path; libzmq drops at the high-water mark)."""
    from serl_amd.transport import zmtp
    pub = zmtp.Socket(zmtp.PUB)
    port, _ = _free_ports()
    pub.bind(f"tcp://127.0.0.1:{port}")
    raw = socket.create_connection(("127.0.0.1", port))          # a hand-made SUB that subscribes and then never reads again
    raw.sendall(zmtp._GREETING)
    raw.sendall(zmtp._ready(zmtp.SUB))
    raw.sendall(zmtp._frame(b"\x01"))                             # ZMTP 3.0 subscription message: subscribe to everything
    deadline = time.time() + 5
    while time.time() < deadline and not any(p.subs for p in pub._peers):
        time.sleep(0.02)
    assert any(p.subs for p in pub._peers), "the subscription did not arrive"
    payload = b"x" * (1 << 20)
    t0 = time.time()
    for _ in range(64):                                           # 64 MB into a socket nobody drains
        pub.send(payload)
    dt = time.time() - t0
    assert dt < 5.0, f"publisher blocked for {dt:.1f} s on a stalled subscriber"
    # whatever reached the wire is whole frames: 0xFF-free framing check by parsing everything that can be read
    raw.settimeout(0.5)
    buf = bytearray()
    try:
        while True:
            chunk = raw.recv(1 << 20)
            if not chunk:
                break
            buf += chunk
    except (socket.timeout, OSError):
        pass
    buf = buf[64:]                                                 # the publisher's greeting
    n_msgs, torn = 0, False
    while len(buf) >= 2:
        fl = buf[0]
        size, hdr = (int.from_bytes(buf[1:9], "big"), 9) if fl & 2 else (buf[1], 2)
        if len(buf) < hdr + size:
            torn = True                                            # a partial frame is only legal as the LAST thing on a closed link
            break
        if not fl & 4:
            assert bytes(buf[hdr:hdr + size]) == payload
            n_msgs += 1
        del buf[:hdr + size]
    assert not torn or not any(p.alive for p in pub._peers), "a partial frame was left on a live connection"
    pub.close()
    raw.close()


def test_pub_delivers_large_messages_to_a_slow_but_reading_subscriber():
    """the mid-frame budget was wall-clock from the first byte (50 ms), so a 20 MB parameter publish to a
    subscriber that reads slowly (busy threads, 1 GbE) closed the link and the actor kept a stale policy forever.  The stall
    timer now restarts on every byte of progress: a reader that keeps reading gets every message, however slowly."""
    from serl_amd.transport import zmtp
    pub, sub = zmtp.Socket(zmtp.PUB), zmtp.Socket(zmtp.SUB)
    port, _ = _free_ports()
    pub.bind(f"tcp://127.0.0.1:{port}")
    sub.setsockopt(zmtp.SUBSCRIBE, b"")
    sub.setsockopt(zmtp.RCVTIMEO, 20000)
    # throttle the subscriber's I/O thread: at most 256 KB per read every 2 ms (~100 MB/s, far below loopback speed), so a
    # 20 MB frame takes ~200 ms -- four times the old fixed budget -- while bytes keep flowing
    real_pump = zmtp._Peer.pump

    def slow_pump(self):
        time.sleep(0.002)
        try:
            chunk = self.conn.recv(1 << 18)
            if not chunk:
                self.alive = False
            self.buf += chunk
        except (BlockingIOError, InterruptedError):
            pass
        except OSError:
            self.alive = False
        conn, self.conn = self.conn, _NoRead()
        try:
            real_pump(self)                      # frame parsing only
        finally:
            self.conn = conn

    class _NoRead:
        def recv(self, n):
            raise BlockingIOError

    sub.connect(f"tcp://127.0.0.1:{port}")
    deadline = time.time() + 5
    while time.time() < deadline and not any(p.subs for p in pub._peers):
        time.sleep(0.02)
    assert any(p.subs for p in pub._peers)
    for p in sub._peers:
        p.pump = slow_pump.__get__(p)
    payloads = [bytes([k]) * (20 << 20) for k in range(3)]
    got = []
    rx = threading.Thread(target=lambda: [got.append(sub.recv()) for _ in payloads])
    rx.start()
    for m in payloads:
        pub.send(m)
    rx.join(timeout=60)
    assert [g[:1] for g in got] == [m[:1] for m in payloads] and all(len(g) == 20 << 20 for g in got)
    assert any(p.alive for p in pub._peers), "the publisher gave up on a subscriber that was reading"
    pub.close()
    sub.close()


def test_pub_gives_up_on_a_trickle_reading_subscriber_within_the_message_deadline():
    """the stall timer restarts on every byte of progress, so a peer that reads a few KB inside every stall window
    would hold the publisher (and every peer served after it) for ever.  The message as a whole has a deadline
    (stall + len / min_rate); past it the connection is closed like any mid-frame stall."""
    import socket as pysock
    from serl_amd.transport import zmtp
    lst = pysock.socket()
    lst.bind(("127.0.0.1", 0))
    lst.listen(1)
    cli = pysock.create_connection(lst.getsockname())
    srv, _ = lst.accept()
    srv.setblocking(False)
    srv.setsockopt(pysock.SOL_SOCKET, pysock.SO_SNDBUF, 1 << 16)
    cli.setsockopt(pysock.SOL_SOCKET, pysock.SO_RCVBUF, 1 << 16)
    peer = zmtp._Peer.__new__(zmtp._Peer)
    peer.conn, peer.alive, peer.slock = srv, True, threading.Lock()
    stop = threading.Event()

    def trickle():                      # 4 KB every 50 ms: always "progress" inside a 0.5 s stall window, 80 KB/s overall
        while not stop.is_set():
            try:
                if not cli.recv(4096):
                    return
            except OSError:
                return
            time.sleep(0.05)

    th = threading.Thread(target=trickle, daemon=True)
    th.start()
    t0 = time.time()
    ok = peer.send_or_drop(b"x" * (8 << 20), stall=0.5, min_rate=8e6)    # deadline 0.5 + 1.0 s; the trickle would need ~100 s
    dt = time.time() - t0
    stop.set()
    assert ok is False and not peer.alive and dt < 5.0, (ok, peer.alive, dt)
    cli.close()
    lst.close()


def test_sub_reconnects_after_the_publisher_dropped_it():
    """ADVICE r4 (high, second half): a SUB whose connection was closed must dial again and re-send its subscriptions
    (libzmq does); before, the dead peer was removed and the endpoint never returned to the pending list."""
    from serl_amd.transport import zmtp
    pub, sub = zmtp.Socket(zmtp.PUB), zmtp.Socket(zmtp.SUB)
    port, _ = _free_ports()
    pub.bind(f"tcp://127.0.0.1:{port}")
    sub.setsockopt(zmtp.SUBSCRIBE, b"")
    sub.setsockopt(zmtp.RCVTIMEO, 5000)
    sub.connect(f"tcp://127.0.0.1:{port}")
    deadline = time.time() + 5
    while time.time() < deadline and not any(p.subs for p in pub._peers):
        time.sleep(0.02)
    pub.send(b"one")
    assert sub.recv() == b"one"
    for p in list(pub._peers):           # the publisher gives the subscriber up (what a mid-frame stall does)
        p.close()
    deadline = time.time() + 10
    while time.time() < deadline and not any(p.alive and p.subs for p in pub._peers):
        time.sleep(0.05)
    assert any(p.alive and p.subs for p in pub._peers), "the subscriber did not come back"
    pub.send(b"two")
    assert sub.recv() == b"two"
    pub.close()
    sub.close()


def test_truncated_lz4_frame_is_an_error():
    from serl_amd.transport import lz4frame
    if lz4frame._load() is None:
        import pytest
        pytest.skip("liblz4 is not loadable here")
    frame = lz4frame.compress(bytes(range(256)) * 400)
    assert lz4frame.decompress(frame) == bytes(range(256)) * 400
    import pytest
    with pytest.raises(ValueError):
        lz4frame.decompress(frame[:len(frame) // 2])
