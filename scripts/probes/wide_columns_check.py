import os, sys, numpy as np, torch, threading
sys.path.insert(0, os.getcwd())
import chalametpir_amd as cp
from oracle import oracle as orc
rng = np.random.default_rng(5)
dev = cp.Device(0); stream = torch.cuda.current_stream(); bad = 0
for b, N, C in ((9, 2048 + 3, 12288), (9, 1536, 12289), (9, 1024, 14593), (10, 700, 20000), (8, 4096, 16385), (9, 600, 40000)):
    D = rng.integers(0, 1 << b, size=(N, C), dtype=np.uint64).astype(np.uint32)
    srv = cp.Server.from_compressed(orc.row_wise_compress(orc.transpose(D), b), N, b, device=dev)
    dtc = orc.row_wise_compress(orc.transpose(D), b)
    qs = rng.integers(0, 1 << 32, size=(26, N), dtype=np.uint64).astype(np.uint32)
    want = np.stack([orc.row_vector_x_compressed_transposed_matrix(q, dtc, N, b)[0] for q in qs])
    q_dev = torch.from_numpy(qs.view(np.int32)).cuda(); r_dev = torch.full((26, C), -1, dtype=torch.int32, device="cuda")
    srv.respond_batch_device(q_dev, 26, r_dev, stream=stream); torch.cuda.synchronize()
    ok_dev = np.array_equal(r_dev.cpu().numpy().view(np.uint32), want)
    ok_host = all(np.array_equal(srv.respond_array(qs[i]), want[i]) for i in range(3))
    pin = cp.PinnedArray(N); pin.array[:] = qs[4]
    ok_pin = np.array_equal(srv.respond_array(pin.array), want[4])
    res = [None] * 6
    def work(k): res[k] = np.array_equal(srv.respond_array(qs[10 + k]), want[10 + k])
    ts = [threading.Thread(target=work, args=(k,)) for k in range(6)]; [t.start() for t in ts]; [t.join() for t in ts]
    print(f"b={b} N={N} C={C}: device batch {ok_dev}, lone host {ok_host}, page-locked {ok_pin}, 6 concurrent {all(res)}; served {srv.host_path_counts()}", flush=True)
    bad += not (ok_dev and ok_host and ok_pin and all(res))
    pin.close(); srv.close()
print("wide-C check:", bad, "bad"); sys.exit(1 if bad else 0)
