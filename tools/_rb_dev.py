import sys, os, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import oraclelib
from mercury_amd import RxPhy
cfg, W = 8, 1024
orc = oraclelib.Oracle(cfg)
n = orc.buffer_samples()
rng = np.random.default_rng(1)
wins = rng.standard_normal((W, n)) * 0.01
for w in range(W):
    pl = rng.integers(0, 256, orc.payload_bytes)
    pb = orc.tx_passband(orc.payload_to_bits(pl))
    d = int(rng.integers(5 * 1088, n - pb.size - 5 * 1088))
    wins[w, d: d + pb.size] += pb
rx = RxPhy(cfg, max_batch=W)
dev = torch.from_numpy(wins).to('cuda:0')
torch.cuda.synchronize()
for i in range(4):
    t0 = time.perf_counter()
    r = rx.receive_byte_dev(dev.data_ptr(), W, oraclelib.CARRIER)
    print('ms', (time.perf_counter() - t0) * 1e3, int(r['stats']['message_decoded'].sum()), flush=True)
