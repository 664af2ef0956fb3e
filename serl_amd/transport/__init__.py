"""Actor <-> learner endpoint with agentlace's surface (the reference's un-vendored transport dependency, pinned at git
cf2c337 by serl_launcher/setup.py:16; call sites: utils/launcher.py:171-177, data/data_store.py:83-144,
examples/async_drq_sim/async_drq_sim.py:95-108,161-171,202-229,297).

    TrainerConfig, TrainerServer, TrainerClient, DataStoreBase, QueuedDataStore

Message contract (SURVEY.md 5.8 -- reconstructed from the call sites; agentlace's source is not available here.  What IS
verified: the socket layer -- this package's ZMTP 3.0 implementation talks to a real libzmq in both directions for
REQ/REP and PUB/SUB (tests/test_zmtp_interop.py) -- and the LZ4 frame codec (liblz4).  What is ASSUMED about
agentlace@cf2c337: the dict schema below, `pickle.dumps` + `lz4.frame.compress` as the payload codec, single-frame
messages, and ports 5488 / 5489 -- INTEGRATION.md lists every assumed byte):
  * REQ/REP channel (config.port_number): dict messages
      {"type": "handshake", "config_hash": ...}                                  -> {"success": bool, ...}
      {"type": "datastore", "store_name": name, "payload": [transition, ...]}    -> inserted into the registered store
      {"type": <one of config.request_types>, "payload": dict}                   -> request_callback(type, payload)
  * PUB/SUB channel (config.broadcast_port): publish_network(params) -> every client's recv_network_callback(params)
  * every message is pickle.dumps + lz4.frame.compress

Two transports behind the same classes:
  * "zmq" (default): TCP sockets speaking ZeroMQ's wire protocol -- through pyzmq when it is installed, else through
    serl_amd/transport/zmtp.py (pure Python, same API subset); frames are real LZ4 frames (the `lz4` package, else liblz4
    via ctypes).  tests/test_transport_tcp_cpu.py runs the whole actor <-> learner flow over it;
  * "loopback": in-process registry keyed by port with the same message flow, framing and threading (the server's
    handler thread calls store.insert() concurrently with the learner thread -- the contract of data_store.py:96-136).
"""
from .endpoint import DataStoreBase, QueuedDataStore, TrainerClient, TrainerConfig, TrainerServer, make_trainer_config  # noqa: F401
