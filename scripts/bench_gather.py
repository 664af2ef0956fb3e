import itertools, time, numpy as np, torch, sys
sys.path.insert(0, "/root/repo")
from tests.helpers import make_spaces
from serl_amd.agents.batch import DeviceBatch
from serl_amd.data.data_store import MemoryEfficientReplayBufferDataStore, gather_crop
from serl_amd.utils.synthetic import transition_stream
keys=("front","wrist")
osp, asp = make_spaces(keys,128,128,3,1,24,6)
rb = MemoryEfficientReplayBufferDataStore(osp, asp, 20000, image_keys=keys); rb.seed(0)
t=time.time()
for tr in itertools.islice(transition_stream(keys, seed=1234), 5000): rb.insert(tr)
print("insert 5000:", time.time()-t, "s")
B=256; out=DeviceBatch(B,2,128,128,3,24,6,0)
rng=np.random.default_rng(0)
idxs=[rb.sample_indices(B) for _ in range(50)]
co=rng.integers(0,9,(B,2)).astype(np.int32); cn=rng.integers(0,9,(B,2)).astype(np.int32)
for i in range(5): gather_crop([(rb,idxs[i])],co,cn,out)
torch.cuda.synchronize()
e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(50): gather_crop([(rb,idxs[i])],co,cn,out)
e1.record(); torch.cuda.synchronize()
ms=e0.elapsed_time(e1)/50
print(f"gather_crop: {ms*1e3:.1f} us/call  -> {100.72e6/ (ms*1e-3)/1e12:.2f} TB/s algorithmic")
t=time.time()
for i in range(50): rb.sample_indices(B)
print("sample_indices us:", (time.time()-t)/50*1e6)
