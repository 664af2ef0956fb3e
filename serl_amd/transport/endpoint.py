"""See serl_amd/transport/__init__.py."""
from __future__ import annotations

import hashlib
import pickle
import queue
import threading
import time
import zlib
from collections import deque
from dataclasses import dataclass, field
from typing import Any, Callable, Dict, List, Optional

# Sockets: pyzmq when it is installed, else this package's own implementation of the same wire protocol (ZMTP 3.0 over
# TCP, serl_amd/transport/zmtp.py -- verified against a real libzmq in tests/test_zmtp_interop.py) behind the same API
# subset.  Framing: the `lz4` package when installed, else liblz4 through ctypes (serl_amd/transport/lz4frame.py: real LZ4
# frames), else zlib with a marker byte (only a peer running this package can read that).
try:
    import zmq as _zmq
    ZMQ_BACKEND = "pyzmq"
except Exception:  # noqa: BLE001
    from . import zmtp as _zmq
    ZMQ_BACKEND = "serl_amd.transport.zmtp"
try:
    import lz4.frame as _lz4
    LZ4_BACKEND = "lz4"
except Exception:  # noqa: BLE001
    from . import lz4frame as _lz4
    LZ4_BACKEND = "liblz4 (ctypes)"
    if not _lz4.available():
        _lz4, LZ4_BACKEND = None, "zlib (no lz4 on this host)"


# ------------------------------------------------------------------------------------------------- data stores
class DataStoreBase:
    """agentlace.data.data_store.DataStoreBase (abstract): what TrainerServer / TrainerClient need from a store."""

    def __init__(self, capacity: int):
        self.capacity = capacity

    def insert(self, data):
        raise NotImplementedError

    def batch_insert(self, batch: List[Any]):
        for d in batch:
            self.insert(d)

    def latest_data_id(self):
        raise NotImplementedError

    def get_latest_data(self, from_id: int):
        raise NotImplementedError

    def __len__(self):
        raise NotImplementedError


class QueuedDataStore(DataStoreBase):
    """Actor-side FIFO of transitions waiting to be shipped (agentlace.data.data_store.QueuedDataStore)."""

    def __init__(self, capacity: int):
        super().__init__(capacity)
        self._q = deque(maxlen=capacity)
        self._latest = -1
        self._lock = threading.Lock()

    def insert(self, data):
        with self._lock:
            self._latest += 1
            self._q.append((self._latest, data))

    def latest_data_id(self):
        return self._latest

    def get_latest_data(self, from_id: int):
        with self._lock:
            return [d for i, d in self._q if i > from_id]

    def latest_since(self, from_id: int):
        """(id of the newest returned item, items newer than from_id) taken atomically, so that a concurrent insert is
        neither shipped twice nor lost by TrainerClient.update."""
        with self._lock:
            items = [(i, d) for i, d in self._q if i > from_id]
        return (items[-1][0] if items else from_id), [d for _, d in items]

    def __len__(self):
        return len(self._q)


# ------------------------------------------------------------------------------------------------- config / framing
@dataclass
class TrainerConfig:
    port_number: int = 5488
    broadcast_port: int = 5489
    request_types: List[str] = field(default_factory=list)
    rate_limit: Optional[int] = None
    version: str = "0.0.2"
    experimental_pipeline_port: Optional[int] = None

    def hash(self) -> str:
        return hashlib.sha1(repr((self.port_number, self.broadcast_port, sorted(self.request_types), self.version)).encode()).hexdigest()


def make_trainer_config(port_number: int = 5488, broadcast_port: int = 5489):
    """utils/launcher.py:171-177"""
    return TrainerConfig(port_number=port_number, broadcast_port=broadcast_port, request_types=["send-stats"])


def encode(msg) -> bytes:
    raw = pickle.dumps(msg, protocol=pickle.HIGHEST_PROTOCOL)
    return _lz4.compress(raw) if _lz4 is not None else b"Z" + zlib.compress(raw, 1)


def decode(frame: bytes):
    if _lz4 is not None and not frame.startswith(b"Z"):
        return pickle.loads(_lz4.decompress(frame))
    return pickle.loads(zlib.decompress(frame[1:]))


# ------------------------------------------------------------------------------------------------- loopback fabric
class _Loopback:
    """In-process stand-in for the two ZeroMQ channels: per port a request queue served by the server's handler thread,
    per broadcast port a list of subscriber queues."""
    lock = threading.Lock()
    req: Dict[int, "queue.Queue"] = {}
    subs: Dict[int, List["queue.Queue"]] = {}

    @classmethod
    def bind(cls, port):
        with cls.lock:
            if port in cls.req:
                raise OSError(f"port {port} already bound by another TrainerServer in this process")
            cls.req[port] = queue.Queue()
            return cls.req[port]

    @classmethod
    def unbind(cls, port, bport):
        with cls.lock:
            cls.req.pop(port, None)
            cls.subs.pop(bport, None)

    @classmethod
    def connect(cls, port, wait, timeout=30.0):
        t0 = time.time()
        while True:
            with cls.lock:
                if port in cls.req:
                    return cls.req[port]
            if not wait or time.time() - t0 > timeout:
                raise ConnectionError(f"no TrainerServer on port {port}")
            time.sleep(0.01)

    @classmethod
    def subscribe(cls, bport):
        q = queue.Queue()
        with cls.lock:
            cls.subs.setdefault(bport, []).append(q)
        return q

    @classmethod
    def publish(cls, bport, frame):
        with cls.lock:
            subs = list(cls.subs.get(bport, []))
        for q in subs:
            q.put(frame)


# ------------------------------------------------------------------------------------------------- server
class TrainerServer:
    """Learner-side endpoint: `register_data_store(name, store)`, `start(threaded=True)`, `publish_network(params)`,
    `stop()`; `request_callback(type, payload) -> dict` answers the custom request types."""

    def __init__(self, config: TrainerConfig, request_callback: Optional[Callable[[str, dict], dict]] = None,
                 transport: Optional[str] = None, bind_ip: str = "127.0.0.1"):
        """transport: "zmq" (TCP sockets; default) or "loopback" (in-process queues).  bind_ip: interface the two sockets
        listen on -- "127.0.0.1" by default; "*" (all interfaces, what agentlace does) is an explicit opt-in.  TRUST MODEL: messages are pickles -- whoever can reach the port can
        execute code in the learner process (upstream agentlace has the same property); bind to a private interface or
        "127.0.0.1" unless the network is trusted."""
        self.config, self.request_callback = config, request_callback
        self.transport = transport or "zmq"
        if self.transport not in ("zmq", "loopback"):
            raise ValueError(f"unknown transport {self.transport!r}")
        self.bind_ip = bind_ip
        self.data_stores: Dict[str, DataStoreBase] = {}
        self._thread, self._stop = None, threading.Event()
        self.stats = {"datastore_msgs": 0, "transitions": 0, "requests": 0, "published": 0}

    def register_data_store(self, name: str, data_store: DataStoreBase):
        self.data_stores[name] = data_store

    # -- message handler (runs on the server thread)
    def handle(self, msg: dict) -> dict:
        t = msg.get("type")
        if t == "handshake":
            ok = msg.get("config_hash") == self.config.hash()
            return {"success": ok, "message": "" if ok else "TrainerConfig mismatch between client and server"}
        if t == "datastore":
            store = self.data_stores.get(msg.get("store_name"))
            if store is None:
                return {"success": False, "message": f"unknown data store {msg.get('store_name')!r}"}
            payload = msg["payload"]
            store.batch_insert(payload) if hasattr(store, "batch_insert") else [store.insert(d) for d in payload]
            self.stats["datastore_msgs"] += 1
            self.stats["transitions"] += len(payload)
            return {"success": True}
        if t in self.config.request_types:
            self.stats["requests"] += 1
            out = self.request_callback(t, msg.get("payload")) if self.request_callback else {}
            return {"success": True, "payload": out}
        return {"success": False, "message": f"Invalid request type: {t}"}

    def _serve_loopback(self):
        while not self._stop.is_set():
            try:
                frame, reply = self._inbox.get(timeout=0.05)
            except queue.Empty:
                continue
            try:
                reply.put(encode(self.handle(decode(frame))))
            except Exception as e:  # noqa: BLE001
                reply.put(encode({"success": False, "message": repr(e)}))

    def _serve_zmq(self):
        rep = self._rep
        poller = _zmq.Poller()
        poller.register(rep, _zmq.POLLIN)
        while not self._stop.is_set():
            if dict(poller.poll(50)).get(rep):
                frame = rep.recv()
                # a REP socket that received a request MUST answer it, or every actor blocks in recv forever: a malformed
                # frame or a failing handler becomes an error reply, the server thread keeps running
                try:
                    reply = self.handle(decode(frame))
                except Exception as e:  # noqa: BLE001
                    self.stats["errors"] = self.stats.get("errors", 0) + 1
                    reply = {"success": False, "message": repr(e)}
                rep.send(encode(reply))
        rep.close(0)

    def start(self, threaded: bool = False):
        if self.transport == "loopback":
            self._inbox = _Loopback.bind(self.config.port_number)
            target = self._serve_loopback
        else:
            ctx = _zmq.Context.instance()
            self._rep = ctx.socket(_zmq.REP)
            self._rep.bind(f"tcp://{self.bind_ip}:{self.config.port_number}")
            self._pub = ctx.socket(_zmq.PUB)
            self._pub.bind(f"tcp://{self.bind_ip}:{self.config.broadcast_port}")
            target = self._serve_zmq
        if threaded:
            self._thread = threading.Thread(target=target, name="TrainerServer", daemon=True)
            self._thread.start()
        else:
            target()

    def publish_network(self, params: dict):
        """Broadcast the learner's parameters (a flax-layout tree of numpy arrays) to every actor."""
        frame = encode(params)
        if self.transport == "loopback":
            _Loopback.publish(self.config.broadcast_port, frame)
        else:
            self._pub.send(frame)
        self.stats["published"] += 1

    def stop(self):
        self._stop.set()
        if self._thread is not None:
            self._thread.join(timeout=5)
        if self.transport == "loopback":
            _Loopback.unbind(self.config.port_number, self.config.broadcast_port)
        elif getattr(self, "_pub", None) is not None:
            self._pub.close(0)
            self._pub = None


# ------------------------------------------------------------------------------------------------- client
class TrainerClient:
    """Actor-side endpoint: `update()` ships the new transitions of `data_store`, `request(type, payload)`,
    `recv_network_callback(fn)` delivers every published network to fn on a subscriber thread."""

    def __init__(self, name: str, server_ip: str, config: TrainerConfig, data_store: Optional[DataStoreBase] = None,
                 log_level=None, wait_for_server: bool = False, transport: Optional[str] = None, timeout: float = 30.0):
        self.name, self.config, self.data_store = name, config, data_store
        self.transport = transport or "zmq"
        self.timeout = timeout
        self.last_sync_data_id = -1
        self._stop = threading.Event()
        self._sub_thread = None
        if self.transport == "loopback":
            self._outbox = _Loopback.connect(config.port_number, wait_for_server, timeout)
        else:
            ctx = _zmq.Context.instance()
            self._req = ctx.socket(_zmq.REQ)
            self._req.setsockopt(_zmq.RCVTIMEO, int(timeout * 1000))
            self._req.setsockopt(_zmq.SNDTIMEO, int(timeout * 1000))
            self._req.connect(f"tcp://{server_ip}:{config.port_number}")
            self._ip = server_ip
        res = self._send({"type": "handshake", "config_hash": config.hash()})
        if not res or not res.get("success"):
            raise ConnectionError(f"handshake with the trainer failed: {res and res.get('message')}")

    def _send(self, msg: dict) -> Optional[dict]:
        frame = encode(msg)
        if self.transport == "loopback":
            reply: "queue.Queue" = queue.Queue()
            self._outbox.put((frame, reply))
            try:
                return decode(reply.get(timeout=self.timeout))
            except queue.Empty:
                return None
        try:
            self._req.send(frame)
            return decode(self._req.recv())
        except Exception:  # noqa: BLE001  (timeout / peer gone): a REQ socket that missed its reply cannot be reused
            self._req.close(0)
            self._req = _zmq.Context.instance().socket(_zmq.REQ)
            self._req.setsockopt(_zmq.RCVTIMEO, int(self.timeout * 1000))
            self._req.setsockopt(_zmq.SNDTIMEO, int(self.timeout * 1000))
            self._req.connect(f"tcp://{self._ip}:{self.config.port_number}")
            return None

    def update(self) -> bool:
        """Send everything the local store received since the last successful update."""
        if self.data_store is None:
            return False
        if hasattr(self.data_store, "latest_since"):   # id and items taken atomically: nothing shipped twice, nothing lost
            latest, batch = self.data_store.latest_since(self.last_sync_data_id)
        else:   # a foreign DataStoreBase: read the id FIRST -- a concurrent insert is then re-sent, never dropped
            latest = self.data_store.latest_data_id()
            batch = self.data_store.get_latest_data(self.last_sync_data_id)
        if not batch:
            return True
        res = self._send({"type": "datastore", "store_name": self.name, "payload": batch})
        if res and res.get("success"):
            self.last_sync_data_id = latest
            return True
        return False

    def request(self, type: str, payload: dict) -> Optional[dict]:  # noqa: A002
        res = self._send({"type": type, "payload": payload})
        if res is None or not res.get("success"):
            return None
        return res.get("payload")

    def recv_network_callback(self, callback_fn: Callable[[dict], None]):
        if self.transport == "loopback":
            q = _Loopback.subscribe(self.config.broadcast_port)

            def loop():
                while not self._stop.is_set():
                    try:
                        callback_fn(decode(q.get(timeout=0.05)))
                    except queue.Empty:
                        continue
        else:
            sub = _zmq.Context.instance().socket(_zmq.SUB)
            sub.connect(f"tcp://{self._ip}:{self.config.broadcast_port}")
            sub.setsockopt(_zmq.SUBSCRIBE, b"")

            def loop():
                poller = _zmq.Poller()
                poller.register(sub, _zmq.POLLIN)
                while not self._stop.is_set():
                    if dict(poller.poll(50)).get(sub):
                        callback_fn(decode(sub.recv()))
                sub.close(0)
        self._sub_thread = threading.Thread(target=loop, name="TrainerClient-sub", daemon=True)
        self._sub_thread.start()

    def stop(self):
        self._stop.set()
        if self._sub_thread is not None:
            self._sub_thread.join(timeout=5)
        if self.transport != "loopback" and getattr(self, "_req", None) is not None:
            self._req.close(0)
            self._req = None
