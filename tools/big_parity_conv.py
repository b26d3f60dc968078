"""one-off large differential run of the generic trellis: n noisy blocks per shape (all nine rows of
lower_mac/tetra_conv_enc.c:257-267) with erasures and non-binary bytes, every decoded bit against the oracle
(threads: the oracle call releases the GIL)"""
import sys, os, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch
import osmo_tetra_amd as T, oraclelib as O
import test_gpu_parity as G
eng = T.Engine(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
hs = torch.cuda.current_stream().cuda_stream
for shape in O.PUNCT_SHAPES:
    L, K, mother, pu = shape
    t0 = time.time()
    base = 2000
    t2, t3b = G._conv_batch(shape, base, seed=L + pu)
    rng = np.random.default_rng(K)
    t3 = np.tile(t3b, (n // base, 1))
    t3 ^= ((rng.random(t3.shape) < 0.04) & (t3 < 2)).astype(np.uint8)       # fresh noise on every copy
    d_in = torch.from_numpy(t3.reshape(-1)).cuda()
    d_out = torch.zeros(len(t3) * L, dtype=torch.uint8, device="cuda")
    cv = T.ConvDecoder(eng, pu, mother, K, L)
    cv.execute(d_in.data_ptr(), len(t3), d_out.data_ptr(), hs)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy().reshape(len(t3), L)
    def chunk(r):
        return sum(int((got[i] == O.conv_decode_block(pu, mother, t3[i], L, 0)).all()) for i in r)
    with ThreadPoolExecutor(16) as ex:
        same = sum(ex.map(chunk, [range(a, min(a + 500, len(t3))) for a in range(0, len(t3), 500)]))
    print("shape %s: %d of %d blocks bit-exact, %.1f s" % (shape, same, len(t3), time.time() - t0))
    assert same == len(t3)
    cv.close()
