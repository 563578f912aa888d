import sys, os, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import test_gpu_pipeline as t
try:
    t.test_soak_beside_a_collective_shaped_interferer(torch.device("cuda:0"), "qws_two_slots")
    print("plain python: PASS")
except BaseException as e:
    print("plain python: FAIL", type(e).__name__, str(e)[:300])
print(open("gpurun_out/parity/interferer_soak_qws_two_slots.json").read()[-200:] if os.path.exists("gpurun_out/parity/interferer_soak_qws_two_slots.json") else "no record")
