import ctypes, sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import chalametpir_amd as cp
from oracle import oracle as orc
from _cases import random_db_matrix, random_query
rng = np.random.default_rng(77)
N, C, b = 3 * 1536 + 77, 33, 9
D = random_db_matrix(rng, N, C, b)
device = cp.Device(0)
srv, _ = cp.Server.setup_from_matrix(bytes(range(32)), D, b, device=device)
dtc = orc.row_wise_compress(orc.transpose(D), b)
hip = ctypes.CDLL("libamdhip64.so")
class RawStream:  # a stream of the runtime's own (not one of torch's pool), with the two members the wrappers use
    def __init__(self):
        h = ctypes.c_void_p(); assert hip.hipStreamCreateWithFlags(ctypes.byref(h), 1) == 0; self.cuda_stream = h.value
    def synchronize(self): assert hip.hipStreamSynchronize(ctypes.c_void_p(self.cuda_stream)) == 0
stream = RawStream() if os.environ.get("RAW_STREAM") == "1" else torch.cuda.Stream()
nb = 5
q_dev = torch.zeros((nb, N), dtype=torch.int32, device="cuda")
r1 = torch.zeros(C, dtype=torch.int32, device="cuda")
rb = torch.zeros((nb, C), dtype=torch.int32, device="cuda")
srv.respond_device(q_dev[0], r1, stream=stream); srv.respond_batch_device(q_dev, nb, rb, stream=stream); stream.synchronize()
s_ptr = ctypes.c_void_p(stream.cuda_stream)
if os.environ.get("DUMMY_FIRST") == "1":  # a throw-away graph (one memset) instantiated and launched twice before the real ones
    scratch = torch.zeros(64, dtype=torch.int32, device="cuda"); torch.cuda.synchronize()
    assert hip.hipStreamBeginCapture(s_ptr, 2) == 0
    assert hip.hipMemsetAsync(ctypes.c_void_p(scratch.data_ptr()), 0, 256, s_ptr) == 0
    g0 = ctypes.c_void_p(); assert hip.hipStreamEndCapture(s_ptr, ctypes.byref(g0)) == 0
    i0 = ctypes.c_void_p(); assert hip.hipGraphInstantiate(ctypes.byref(i0), g0, None, None, 0) == 0
    for _ in range(2): assert hip.hipGraphLaunch(i0, s_ptr) == 0
    stream.synchronize()
MODE = int(os.environ.get("CAPTURE_MODE", "2"))
for what in (os.environ.get("ORDER", "one,batch,both").split(",")):
    assert hip.hipStreamBeginCapture(s_ptr, MODE) == 0
    if what in ("one", "both"): srv.respond_device(q_dev[0], r1, stream=stream)
    if what in ("batch", "both"): srv.respond_batch_device(q_dev, nb, rb, stream=stream)
    graph = ctypes.c_void_p(); assert hip.hipStreamEndCapture(s_ptr, ctypes.byref(graph)) == 0
    n = ctypes.c_size_t(0); hip.hipGraphGetNodes(graph, None, ctypes.byref(n)); 
    inst = ctypes.c_void_p(); assert hip.hipGraphInstantiate(ctypes.byref(inst), graph, None, None, 0) == 0
    prev_want = None
    for rep in range(3):
        qs = np.stack([random_query(rng, N) for _ in range(nb)])
        q_dev.copy_(torch.from_numpy(qs.view(np.int32))); r1.fill_(-1); rb.fill_(-1); torch.cuda.synchronize()
        assert hip.hipGraphLaunch(inst, s_ptr) == 0; stream.synchronize()
        want = [orc.row_vector_x_compressed_transposed_matrix(qs[i], dtc, N, b)[0] for i in range(nb)]
        g1 = r1.cpu().numpy().view(np.uint32); gb = rb.cpu().numpy().view(np.uint32)
        # eager for comparison
        e1 = torch.zeros_like(r1); eb = torch.zeros_like(rb)
        srv.respond_device(q_dev[0], e1, stream=stream); srv.respond_batch_device(q_dev, nb, eb, stream=stream); stream.synchronize()
        print(what, "nodes", n.value, "rep", rep, "one ok", np.array_equal(g1, want[0]) if what != "batch" else None, "batch ok", np.array_equal(gb, np.stack(want)) if what != "one" else None,
              "eager one ok", np.array_equal(e1.cpu().numpy().view(np.uint32), want[0]), "eager batch ok", np.array_equal(eb.cpu().numpy().view(np.uint32), np.stack(want)),
              "diff cols one", int((g1 != want[0]).sum()) if what != "batch" else None,
              "one == want - 1 (memset skipped)", bool(np.array_equal(g1, want[0] - np.uint32(1))) if what != "batch" else None,
              "one == previous launch's answer (stale q)", bool(prev_want is not None and np.array_equal(g1, prev_want)) if what != "batch" else None,
              "one == previous - 1", bool(prev_want is not None and np.array_equal(g1, prev_want - np.uint32(1))) if what != "batch" else None)
        if what != "batch" and not np.array_equal(g1, want[0]):
            print("   got ", g1[:6], "\n   want", want[0][:6], "\n   want - got", (want[0] - g1)[:6])
        prev_want = want[0]
    hip.hipGraphExecDestroy(inst); hip.hipGraphDestroy(graph)
