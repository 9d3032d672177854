import sys, time
sys.path.insert(0, "/root/repo")
import torch, hpcla_amd as hp
capi = hp._capi
capi.load()
dev = "cuda"
s = torch.cuda.current_stream().cuda_stream
k = 16
for n in (4096 * 2048, 4096 * 2047):
    rp = torch.arange(0, n + 1, dtype=torch.int32, device=dev)
    cv = torch.arange(0, n, dtype=torch.int32, device=dev)
    nz = torch.ones(n, dtype=torch.float64, device=dev)
    Bc = torch.rand(k, n, dtype=torch.float64, device=dev)
    Cc = torch.empty(k, n, dtype=torch.float64, device=dev)
    Br = torch.rand(n, k, dtype=torch.float64, device=dev)
    Cr = torch.empty(n, k, dtype=torch.float64, device=dev)
    ROW, COL = capi.LAYOUT_ROW, capi.LAYOUT_COL
    def col():
        capi.call("hpcla_spmm_csr_f64_i32", rp.data_ptr(), cv.data_ptr(), nz.data_ptr(), Bc.data_ptr(), n, COL, Cc.data_ptr(), n, COL, n, n, k, 0, s)
    def row():
        capi.call("hpcla_spmm_csr_f64_i32", rp.data_ptr(), cv.data_ptr(), nz.data_ptr(), Br.data_ptr(), k, ROW, Cr.data_ptr(), k, ROW, n, n, k, 0, s)
    def cpy():
        Cc.copy_(Bc)
    for name, fn in (("identity A, column-major", col), ("identity A, row-major", row), ("torch copy of B", cpy)):
        for _ in range(20): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 30
        byts = n * (12 + 4) + 2 * n * k * 8 if "identity" in name else 2 * n * k * 8
        print(f"n={n} {name:28s} {ms:.4f} ms  {byts / ms / 1e6:.0f} GB/s", flush=True)
    assert torch.equal(Cc, Bc)
