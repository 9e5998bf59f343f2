"""Does running W^T W on a side stream, concurrently with W^T A, hide its two launches?  (NMFk sweep shape, small k.)
python tools/dbg/sidegram.py [k]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pydnmfk_amd.engine import HIP_OPS as ops, new_gram
m, n = 65536, 4096
k = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
A = torch.rand(m, n, device=dev); W = torch.rand(m, k, device=dev); H = torch.rand(k, n, device=dev)
G1, G2 = new_gram(k, dev), new_gram(k, dev)
AtW = torch.empty(k, n, device=dev)
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
ws_side = None


def serial():
    ops.gram_hht(H, G1); ops.aht_update_w(A, H, G1, W, 1e-7)
    ops.gram_wtw(W, G2); ops.wta(A, W, AtW); ops.mu_update_h(H, AtW, G2, 1e-7, False)


def forked():
    ops.gram_hht(H, G1); ops.aht_update_w(A, H, G1, W, 1e-7)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        ops.gram_wtw(W, G2)
    ops.wta(A, W, AtW)
    main.wait_stream(side)
    ops.mu_update_h(H, AtW, G2, 1e-7, False)


def nogram():        # upper bound of folding W^T W into the W^T A kernel: the step without its two launches (wrong G2)
    ops.gram_hht(H, G1); ops.aht_update_w(A, H, G1, W, 1e-7)
    ops.wta(A, W, AtW); ops.mu_update_h(H, AtW, G2, 1e-7, False)


def fusedgram():     # dnmf_wta_gram: W^T W rides in the W^T A kernel (one more MFMA per step in one wave per row chunk)
    ops.gram_hht(H, G1); ops.aht_update_w(A, H, G1, W, 1e-7)
    ops.wta_gram(A, W, AtW, G2); ops.mu_update_h(H, AtW, G2, 1e-7, False)


def t(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(2):
    print("k=%d serial %.4f ms  forked %.4f ms  without W^T W %.4f ms  riding W^T W %.4f ms" % (k, t(serial), t(forked), t(nogram), t(fusedgram)))
