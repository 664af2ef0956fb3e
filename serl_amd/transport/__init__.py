"""Actor <-> learner endpoint with agentlace's surface (the reference's un-vendored transport dependency, pinned at git
cf2c337 by serl_launcher/setup.py:16; call sites: utils/launcher.py:171-177, data/data_store.py:83-144,
examples/async_drq_sim/async_drq_sim.py:95-108,161-171,202-229,297).

    TrainerConfig, TrainerServer, TrainerClient, DataStoreBase, QueuedDataStore

Message contract (SURVEY.md 5.8 -- reconstructed from the call sites; agentlace's source is not available here, so the
byte-level wire format over real ZeroMQ is UNVERIFIED):
  * REQ/REP channel (config.port_number): dict messages
      {"type": "handshake", "config_hash": ...}                                  -> {"success": bool, ...}
      {"type": "datastore", "store_name": name, "payload": [transition, ...]}    -> inserted into the registered store
      {"type": <one of config.request_types>, "payload": dict}                   -> request_callback(type, payload)
  * PUB/SUB channel (config.broadcast_port): publish_network(params) -> every client's recv_network_callback(params)
  * every message is pickle.dumps + lz4.frame.compress

Two transports behind the same classes:
  * "zmq": real sockets, used when pyzmq and lz4 are importable (they are not in this image: not exercised here);
  * "loopback": in-process registry keyed by port with the same message flow, framing and threading (the server's
    handler thread calls store.insert() concurrently with the learner thread -- the contract of data_store.py:96-136);
    frames are pickle + zlib level 1 when lz4 is missing.
"""
from .endpoint import DataStoreBase, QueuedDataStore, TrainerClient, TrainerConfig, TrainerServer, make_trainer_config  # noqa: F401
