"""Pure-Python ZeroMQ wire protocol (ZMTP 3.0, NULL mechanism) for the four socket types agentlace uses.

agentlace (pinned at cf2c337 by serl_launcher/setup.py:16; call sites utils/launcher.py:171-177,
examples/async_drq_sim/async_drq_sim.py:95-108,202-229) talks REQ/REP (datastore + requests) and PUB/SUB (network
broadcast) over pyzmq.  pyzmq is not installable in this image, so the learner endpoint could only ever run in-process.
This module speaks the PUBLIC wire protocol itself (ZMTP 3.0, rfc.zeromq.org/spec/23; REQ/REP envelope = spec 28,
PUB/SUB subscriptions = spec 29) over plain TCP sockets, with the small part of pyzmq's API that
serl_amd/transport/endpoint.py uses, so that (a) the endpoint's real-socket code path executes in the CPU test-suite
and (b) a peer running real libzmq (the untouched actor) can connect to it.  (b) is checked against a real libzmq when
one is loadable (tests/test_zmtp_interop.py; skipped otherwise).

Wire format implemented
  greeting   : FF 00*8 7F | 03 00 | "NULL" + 16 x 00 | as-server 00 | 31 x 00                      (64 bytes)
  handshake  : READY command, properties Socket-Type (REQ | REP | PUB | SUB) [+ Identity ""]
  frames     : flags (bit0 MORE, bit1 LONG, bit2 COMMAND) | size (1 byte, or 8 bytes big-endian when LONG) | body
  REQ -> REP : [empty delimiter frame (MORE)] [body]; the reply carries the same envelope back
  SUB -> PUB : message frame 01 <prefix> (subscribe) / 00 <prefix> (cancel) -- the ZMTP 3.0 form; a 3.1 peer's
               SUBSCRIBE / CANCEL command frames are understood too
  PING       : answered with PONG (3.1 peers with heartbeats enabled)
Not implemented: security mechanisms other than NULL, ROUTER/DEALER, multipart application messages (agentlace sends
single-frame messages).  A connection this side dialled is dialled again when it breaks (a SUB re-sends its subscriptions).
"""
from __future__ import annotations

import select
import socket
import struct
import threading
import time
from typing import Dict, List, Optional, Tuple

REQ, REP, PUB, SUB = 3, 4, 1, 2          # pyzmq's constants
POLLIN = 1
SUBSCRIBE, UNSUBSCRIBE, LINGER, RCVTIMEO, SNDTIMEO = 6, 7, 17, 27, 28
_NAMES = {REQ: b"REQ", REP: b"REP", PUB: b"PUB", SUB: b"SUB"}
_PEER_OK = {REQ: (b"REP", b"ROUTER"), REP: (b"REQ", b"DEALER"), PUB: (b"SUB", b"XSUB"), SUB: (b"PUB", b"XPUB")}
_GREETING = b"\xff" + b"\x00" * 8 + b"\x7f" + b"\x03\x00" + b"NULL".ljust(20, b"\x00") + b"\x00" + b"\x00" * 31


class ZMTPError(OSError):
    pass


class Again(ZMTPError):
    """recv / send timed out (pyzmq: zmq.Again)."""


def _frame(body: bytes, more: bool = False, command: bool = False) -> bytes:
    flags = (1 if more else 0) | (4 if command else 0)
    if len(body) > 255:
        return bytes([flags | 2]) + struct.pack(">Q", len(body)) + body
    return bytes([flags, len(body)]) + body


def _ready(sock_type: int) -> bytes:
    def prop(name: bytes, value: bytes) -> bytes:
        return bytes([len(name)]) + name + struct.pack(">I", len(value)) + value
    body = b"\x05READY" + prop(b"Socket-Type", _NAMES[sock_type])
    if sock_type in (REQ,):
        body += prop(b"Identity", b"")
    return _frame(body, command=True)


class _Peer:
    """One established TCP connection: greeting + READY done, inbound byte buffer, parsed frames."""

    def __init__(self, conn: socket.socket, sock_type: int, timeout: float):
        self.conn, self.buf, self.subs = conn, bytearray(), []   # subs: prefixes (PUB side)
        self.frames: List[Tuple[int, bytes]] = []
        self.alive = True
        self.slock = threading.Lock()      # the application thread and the I/O thread (PONG) both write
        conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
        conn.settimeout(timeout)
        conn.sendall(_GREETING)
        g = self._read_exact(64)
        if g[0] != 0xFF or g[9] != 0x7F or g[10] < 3:
            raise ZMTPError(f"peer does not speak ZMTP 3.x (greeting {bytes(g[:12]).hex()})")
        if bytes(g[12:32]).rstrip(b"\x00") != b"NULL":
            raise ZMTPError(f"peer demands security mechanism {bytes(g[12:32]).rstrip(bytes(1))!r}; only NULL is implemented")
        conn.sendall(_ready(sock_type))
        while True:                       # the peer's READY
            fl, body = self._read_frame_blocking()
            if not body:
                raise ZMTPError("empty command frame during the handshake")
            if fl & 4 and body[1:1 + body[0]] == b"READY":
                props, p = {}, 1 + body[0]
                while p < len(body):
                    n = body[p]; name = bytes(body[p + 1:p + 1 + n]); p += 1 + n
                    (vl,) = struct.unpack(">I", body[p:p + 4]); props[name.lower()] = bytes(body[p + 4:p + 4 + vl]); p += 4 + vl
                peer_type = props.get(b"socket-type", b"?")
                if peer_type not in _PEER_OK[sock_type]:
                    raise ZMTPError(f"{_NAMES[sock_type].decode()} socket cannot talk to a {peer_type.decode()} peer")
                break
            if fl & 4 and body[1:1 + body[0]] == b"ERROR":
                raise ZMTPError(f"peer refused the handshake: {bytes(body[7:])!r}")
        conn.setblocking(False)

    def _read_exact(self, n: int) -> bytearray:
        out = bytearray()
        while len(out) < n:
            chunk = self.conn.recv(n - len(out))
            if not chunk:
                raise ZMTPError("connection closed during the handshake")
            out += chunk
        return out

    def _read_frame_blocking(self) -> Tuple[int, bytes]:
        fl = self._read_exact(1)[0]
        size = struct.unpack(">Q", self._read_exact(8))[0] if fl & 2 else self._read_exact(1)[0]
        return fl, bytes(self._read_exact(size))

    def pump(self) -> None:
        """Move whatever the socket has into the frame list (non-blocking)."""
        try:
            while True:
                chunk = self.conn.recv(1 << 20)
                if not chunk:
                    self.alive = False
                    break
                self.buf += chunk
        except (BlockingIOError, InterruptedError):
            pass
        except OSError:
            self.alive = False
        while True:
            if len(self.buf) < 2:
                return
            fl = self.buf[0]
            if fl & 2:
                if len(self.buf) < 9:
                    return
                size, hdr = struct.unpack(">Q", self.buf[1:9])[0], 9
            else:
                size, hdr = self.buf[1], 2
            if len(self.buf) < hdr + size:
                return
            self.frames.append((fl, bytes(self.buf[hdr:hdr + size])))
            del self.buf[:hdr + size]

    def send(self, data: bytes, timeout: float) -> None:
        with self.slock:
            self._send(data, timeout)

    def send_or_drop(self, data: bytes, stall: float = 1.0, min_rate: float = 1e6) -> bool:
        """PUB semantics (libzmq never blocks a publisher: it drops WHOLE messages at the high-water mark): if the socket is
        not writable right now the message is dropped and the stream stays intact.  Once the first byte of a frame is on the
        wire the rest must follow; the connection is given up only when the subscriber accepts NO byte for `stall` seconds
        (the timer restarts on every byte of progress, so a slow link or a busy subscriber that keeps reading -- 20 MB of
        parameters over 1 GbE take 160 ms -- is never cut off); a peer that stopped reading mid-frame cannot be resynchronised,
        so its connection is closed and the SUB side reconnects (Socket._pump).  The whole message also has a deadline,
        `stall + len(data) / min_rate` seconds (1 MB/s: 20 MB of parameters get 21 s): a subscriber that trickle-reads a few bytes
        inside every stall window would otherwise hold this publisher -- and every peer served after it -- for ever (ADVICE r5;
        libzmq's PUB never blocks at all).  -> True if the message went out."""
        with self.slock:
            view, started, last = memoryview(data), False, 0.0
            deadline = time.time() + stall + len(data) / max(min_rate, 1.0)
            while len(view):
                try:
                    n = self.conn.send(view)
                    view = view[n:]
                    if n:
                        started, last = True, time.time()
                except (BlockingIOError, InterruptedError):
                    if not started:
                        return False                 # nothing sent yet: drop the whole message
                    if time.time() - last > stall or time.time() > deadline:
                        self.close()                 # mid-frame and no (or hopelessly slow) progress: never leave a partial frame behind
                        return False
                    select.select([], [self.conn], [], 0.02)
                except OSError:
                    self.close()
                    return False
            return True

    def _send(self, data: bytes, timeout: float) -> None:
        view, t0 = memoryview(data), time.time()
        while len(view):
            try:
                n = self.conn.send(view)
                view = view[n:]
            except (BlockingIOError, InterruptedError):
                if time.time() - t0 > timeout:
                    raise Again("send timed out")
                select.select([], [self.conn], [], 0.05)
            except OSError as e:
                self.alive = False
                raise ZMTPError(f"peer went away: {e}") from e

    def close(self):
        self.alive = False
        try:
            self.conn.close()
        except OSError:
            pass


class Socket:
    def __init__(self, sock_type: int):
        if sock_type not in _NAMES:
            raise ValueError(f"socket type {sock_type} not implemented (REQ, REP, PUB, SUB)")
        self.type = sock_type
        self._listener: Optional[socket.socket] = None
        self._peers: List[_Peer] = []
        self._pending: List[Tuple[str, int]] = []        # endpoints to (re)connect lazily, like libzmq does
        self._subs: List[bytes] = []                     # SUB: our subscriptions
        self._inbox: List[Tuple[_Peer, bytes]] = []      # complete application messages
        self._reply_to: Optional[_Peer] = None           # REP: the peer whose request is being answered
        self._awaiting_reply = False                     # REQ state machine
        self._lock = threading.RLock()
        self._cv = threading.Condition(self._lock)       # new message / new peer
        self._io: Optional[threading.Thread] = None      # what libzmq's I/O thread does: accept, connect, handshake, read
        self.rcvtimeo = self.sndtimeo = -1               # ms, -1 = block
        self.closed = False
        self.port = None

    # ---- pyzmq surface -----------------------------------------------------------------------------------------------
    def bind(self, endpoint: str):
        host, port = _parse(endpoint)
        ls = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        ls.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        ls.bind(("0.0.0.0" if host == "*" else host, port))
        ls.listen(64)
        ls.setblocking(False)
        self._listener, self.port = ls, ls.getsockname()[1]
        self._start_io()

    def connect(self, endpoint: str):
        with self._lock:
            self._pending.append(_parse(endpoint))
        self._start_io()

    def _start_io(self):
        if self._io is None:
            self._io = threading.Thread(target=self._io_loop, name="zmtp-io", daemon=True)
            self._io.start()

    def _io_loop(self):
        while not self.closed:
            try:
                self._pump(0.02)
            except Exception:      # noqa: BLE001 -- a broken peer must not kill the socket's I/O
                time.sleep(0.01)

    def setsockopt(self, opt: int, value):
        if opt == SUBSCRIBE:
            with self._lock:
                self._subs.append(bytes(value))
                peers = list(self._peers)
            for p in peers:
                p.send(_frame(b"\x01" + bytes(value)), 5.0)
        elif opt == UNSUBSCRIBE:
            if bytes(value) in self._subs:
                self._subs.remove(bytes(value))
            for p in self._peers:
                p.send(_frame(b"\x00" + bytes(value)), 5.0)
        elif opt == RCVTIMEO:
            self.rcvtimeo = int(value)
        elif opt == SNDTIMEO:
            self.sndtimeo = int(value)
        elif opt == LINGER:
            pass
        else:
            raise ValueError(f"socket option {opt} not implemented")

    def send(self, data: bytes, flags: int = 0):
        timeout = 30.0 if self.sndtimeo < 0 else self.sndtimeo / 1000.0
        if self.type == PUB:
            body = _frame(bytes(data))
            with self._lock:
                targets = [p for p in self._peers if p.alive and any(bytes(data).startswith(s) for s in p.subs)]
            for p in targets:
                p.send_or_drop(body)         # never blocks the learner; a stalled subscriber loses whole messages or the link
            return
        if self.type == REQ:
            if self._awaiting_reply:
                raise ZMTPError("REQ socket: send() while a reply is outstanding (EFSM)")
            peer = self._wait_peer(timeout)
            peer.send(_frame(b"", more=True) + _frame(bytes(data)), timeout)
            self._awaiting_reply = True
            return
        if self.type == REP:
            with self._lock:
                if self._reply_to is None:
                    raise ZMTPError("REP socket: send() without a request (EFSM)")
                peer, self._reply_to = self._reply_to, None
            peer.send(_frame(b"", more=True) + _frame(bytes(data)), timeout)
            return
        raise ZMTPError("SUB sockets do not send")

    def recv(self, flags: int = 0) -> bytes:
        deadline = None if self.rcvtimeo < 0 else time.time() + self.rcvtimeo / 1000.0
        with self._cv:
            while True:
                if self.type == REP and self._reply_to is not None:
                    raise ZMTPError("REP socket: recv() before the previous request was answered (EFSM)")
                if self._inbox:
                    peer, body = self._inbox.pop(0)
                    if self.type == REP:
                        self._reply_to = peer
                    if self.type == REQ:
                        self._awaiting_reply = False
                    return body
                if self.closed:
                    raise ZMTPError("socket closed")
                left = 0.05 if deadline is None else min(0.05, deadline - time.time())
                if left <= 0:
                    raise Again("recv timed out")
                self._cv.wait(left)

    def poll(self, timeout_ms: int = 0, flags: int = POLLIN) -> int:
        deadline = time.time() + max(timeout_ms, 0) / 1000.0
        with self._cv:
            while not self._inbox and not self.closed:
                left = deadline - time.time()
                if left <= 0:
                    break
                self._cv.wait(min(left, 0.05))
            return POLLIN if self._inbox else 0

    def close(self, linger: int = 0):
        self.closed = True
        if self._io is not None and self._io is not threading.current_thread():
            self._io.join(timeout=2.0)
        with self._cv:
            self._cv.notify_all()
            for p in self._peers:
                p.close()
            self._peers = []
            if self._listener is not None:
                try:
                    self._listener.close()
                except OSError:
                    pass
                self._listener = None

    # ---- internals ---------------------------------------------------------------------------------------------------
    def _wait_peer(self, timeout: float) -> _Peer:
        deadline = time.time() + timeout
        with self._cv:
            while True:
                live = [p for p in self._peers if p.alive]
                if live:
                    return live[0]
                if time.time() > deadline or self.closed:
                    raise Again("no peer connected")
                self._cv.wait(0.05)

    def _add_peer(self, peer: _Peer) -> None:
        with self._cv:
            subs = list(self._subs)
            self._peers.append(peer)
            self._cv.notify_all()
        if self.type == SUB:
            for s in subs:
                peer.send(_frame(b"\x01" + s), 5.0)

    def _pump(self, wait: float) -> None:
        """One round of the I/O thread: connect, accept (+ handshake, outside the lock), read, dispatch."""
        with self._lock:
            pending = list(self._pending)
        for ep in pending:                          # outgoing connections (retried until the server is up)
            try:
                c = socket.create_connection(ep, timeout=1.0)
            except OSError:
                continue
            try:
                peer = _Peer(c, self.type, 5.0)
            except (ZMTPError, OSError):
                c.close()
                continue
            peer.endpoint = ep                      # a connection WE made: re-made when it breaks (libzmq's reconnect)
            with self._lock:
                self._pending.remove(ep)
            self._add_peer(peer)
        with self._lock:
            ls = self._listener
            rl = [p.conn for p in self._peers if p.alive] + ([ls] if ls is not None else [])
        if not rl:
            time.sleep(min(max(wait, 0.0), 0.05))
            return
        try:
            ready, _, _ = select.select(rl, [], [], wait)
        except (OSError, ValueError):
            ready = rl
        if ls is not None and ls in ready:
            while True:
                try:
                    c, _ = ls.accept()
                except (BlockingIOError, InterruptedError, OSError):
                    break
                try:
                    self._add_peer(_Peer(c, self.type, 1.0))
                except Exception:    # noqa: BLE001 -- silent / malformed / half-open connections: close, never leak the socket
                    c.close()
        with self._cv:
            n0 = len(self._inbox)
            for p in list(self._peers):
                if p.conn in ready or p.buf:
                    p.pump()
                self._dispatch(p)
                if not p.alive:
                    self._peers.remove(p)
                    if self._reply_to is p:
                        self._reply_to = None
                    ep = getattr(p, "endpoint", None)
                    if ep is not None and not self.closed and ep not in self._pending:
                        self._pending.append(ep)    # the connecting side dials again (subscriptions are re-sent by _add_peer)
            if len(self._inbox) != n0:
                self._cv.notify_all()

    def _dispatch(self, p: _Peer) -> None:
        """Turn the peer's parsed frames into application messages / protocol actions."""
        frames, p.frames = p.frames, []
        i = 0
        while i < len(frames):
            fl, body = frames[i]
            if fl & 4:                                            # command frame
                name = body[1:1 + body[0]] if body else b""
                if name == b"PING":
                    ctx = body[1 + 4 + 2:]
                    p.send(_frame(b"\x04PONG" + ctx, command=True), 5.0)
                elif name == b"SUBSCRIBE" and self.type == PUB:
                    p.subs.append(bytes(body[10:]))
                elif name == b"CANCEL" and self.type == PUB and bytes(body[7:]) in p.subs:
                    p.subs.remove(bytes(body[7:]))
                i += 1
                continue
            # gather one (possibly multi-frame) message
            j = i
            while j < len(frames) and frames[j][0] & 1:
                j += 1
            if j >= len(frames):                                  # incomplete: put the tail back
                p.frames = frames[i:] + p.frames
                return
            parts = [b for _, b in frames[i:j + 1]]
            i = j + 1
            if self.type == PUB:                                  # subscription messages from a SUB / XSUB peer
                m = parts[-1]
                if m[:1] == b"\x01":
                    p.subs.append(bytes(m[1:]))
                elif m[:1] == b"\x00" and bytes(m[1:]) in p.subs:
                    p.subs.remove(bytes(m[1:]))
                continue
            if self.type in (REQ, REP):                           # strip the envelope up to the empty delimiter
                if b"" in parts:
                    parts = parts[parts.index(b"") + 1:]
            self._inbox.append((p, b"".join(parts)))


class Poller:
    def __init__(self):
        self._socks: List[Socket] = []

    def register(self, sock: Socket, flags: int = POLLIN):
        if sock not in self._socks:
            self._socks.append(sock)

    def poll(self, timeout_ms: int = 0) -> List[Tuple[Socket, int]]:
        t0 = time.time()
        while True:
            out = [(s, POLLIN) for s in self._socks if s.poll(0)]
            left = timeout_ms / 1000.0 - (time.time() - t0)
            if out or left <= 0:
                return out
            if len(self._socks) == 1:
                self._socks[0].poll(int(min(left, 0.05) * 1000))
            else:
                time.sleep(min(left, 0.002))


class Context:
    _instance = None

    @classmethod
    def instance(cls) -> "Context":
        if cls._instance is None:
            cls._instance = cls()
        return cls._instance

    def socket(self, sock_type: int) -> Socket:
        return Socket(sock_type)


def _parse(endpoint: str) -> Tuple[str, int]:
    if not endpoint.startswith("tcp://"):
        raise ValueError(f"only tcp:// endpoints are implemented (got {endpoint!r})")
    host, _, port = endpoint[6:].rpartition(":")
    return host, int(port)
