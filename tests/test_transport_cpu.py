"""CPU: the agentlace-shaped actor <-> learner endpoint (serl_amd/transport) over its in-process loopback: handshake,
datastore shipping into a registered DataStoreBase from the server thread, custom requests, network broadcast -- the
call pattern of examples/async_drq_sim/async_drq_sim.py:95-108,161-171,202-229,297."""
import threading
import time

import numpy as np
import pytest

from serl_amd.transport import DataStoreBase, QueuedDataStore, TrainerClient, TrainerConfig, TrainerServer, make_trainer_config
from serl_amd.transport.endpoint import decode, encode


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
    return {"observations": {"state": np.full((1, 3), k, np.float32), "front": np.full((1, 4, 4, 3), k % 256, np.uint8)},
            "actions": np.zeros(2, np.float32), "rewards": np.float32(k), "masks": np.float32(1), "dones": False}


def test_framing_roundtrip():
    m = {"type": "datastore", "store_name": "actor_env", "payload": [_tr(3)]}
    out = decode(encode(m))
    assert out["type"] == "datastore" and np.array_equal(out["payload"][0]["observations"]["front"], _tr(3)["observations"]["front"])


def test_actor_learner_message_flow():
    cfg = make_trainer_config(port_number=6488, broadcast_port=6489)
    assert cfg.request_types == ["send-stats"]
    got_stats, nets = [], []
    store = ListStore(1000)
    server = TrainerServer(cfg, request_callback=lambda t, p: got_stats.append((t, p)) or {"ack": len(got_stats)}, transport="loopback")
    server.register_data_store("actor_env", store)
    server.start(threaded=True)
    try:
        local = QueuedDataStore(2000)
        client = TrainerClient("actor_env", "localhost", cfg, local, wait_for_server=True, transport="loopback")
        client.recv_network_callback(lambda p: nets.append(p))
        for k in range(25):
            local.insert(_tr(k))
            if k % 10 == 9:
                assert client.update()
        assert len(store) == 20 and [float(d["rewards"]) for d in store.items] == list(range(20))
        assert client.update() and len(store) == 25 and client.update() and len(store) == 25   # nothing new: no resend
        assert store.threads == {"TrainerServer"}                     # inserts run on the server thread
        assert client.request("send-stats", {"eval": {"return": 1.5}}) == {"ack": 1}
        assert got_stats == [("send-stats", {"eval": {"return": 1.5}})]
        assert client.request("bogus", {}) is None                    # not in request_types
        params = {"modules_actor": {"Dense_0": {"kernel": np.arange(6, dtype=np.float32).reshape(3, 2)}}}
        server.publish_network(params)
        t0 = time.time()
        while not nets and time.time() - t0 < 5:
            time.sleep(0.01)
        assert len(nets) == 1 and np.array_equal(nets[0]["modules_actor"]["Dense_0"]["kernel"], params["modules_actor"]["Dense_0"]["kernel"])
        assert server.stats == {"datastore_msgs": 3, "transitions": 25, "requests": 1, "published": 1}
        client.stop()
    finally:
        server.stop()


def test_config_mismatch_is_refused():
    cfg = TrainerConfig(port_number=6490, broadcast_port=6491, request_types=["send-stats"])
    server = TrainerServer(cfg, transport="loopback")
    server.start(threaded=True)
    try:
        other = TrainerConfig(port_number=6490, broadcast_port=6491, request_types=["something-else"])
        with pytest.raises(ConnectionError, match="handshake"):
            TrainerClient("actor_env", "localhost", other, QueuedDataStore(10), wait_for_server=True, transport="loopback")
        with pytest.raises(ConnectionError):
            TrainerClient("x", "localhost", TrainerConfig(port_number=6599), None, wait_for_server=False, transport="loopback")
    finally:
        server.stop()
