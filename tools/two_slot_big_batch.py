"""Two slots in flight at large batch x long sequence: which cooperative launch reports a failed hand-off (diagnostic)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnpn_sc_amd import ops, synth
from gnnpn_sc_amd.pipeline import ML2PNPipeline, DeviceServices, DeviceBatch, PipelinedRunner
from bench import build_models
dev = torch.device("cuda:0")
T, K, S = 1000, 5, 5000
net, low, high = build_models(T, S, K, dev)
table = synth.make_service_table(T, S, seed=0, degree=32)
svc = DeviceServices.from_table(table, dev)
pipe = ML2PNPipeline(net, low, high, K)
for B in [int(v) for v in (sys.argv[1:] or ["1024", "2048", "4096"])]:
    batch = DeviceBatch.from_problems(synth.make_problem_batch(table, B, seed=1, tasks_per_problem=T), dev)
    runner = PipelinedRunner(pipe, svc, batch, slots=2)
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.time()
        for i in range(4):
            runner.submit()
        for st in runner.streams:
            st.synchronize()
        dt = time.time() - t0
        words = [(int(w.status[0]), int(w.encode()[:4].view(torch.int32).item()), int(w._decode[:4].view(torch.int32).item()),
                  int(w.encode()[4:8].view(torch.int32).item())) for w in runner.workspaces]
        print(f"B={B} rep {rep}: 4 steps {dt*1e3:.0f} ms; per slot (sticky, enc word, dec word, enc fast-path WGs) = {words}", flush=True)
        for w in runner.workspaces:
            w.status.zero_()
    del runner, batch
    torch.cuda.empty_cache()
