import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if GOLDEN not in sys.path:
    sys.path.insert(0, GOLDEN)          # tests/golden/pn_inputs.py (seeded inputs shared with the fixture generators)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def _record_box():
    """Which card this run is on (unique id, partition modes, firmware): gpurun_out/parity/box_info.json — hand-off time-outs
    have been box-dependent (profiles/LOG_r01_r04.md section 13.3), so every GPU run of the suite says where it ran."""
    import json
    import subprocess
    info = {}
    try:
        out = subprocess.run(["rocm-smi", "--showuniqueid", "--showmemorypartition", "--showcomputepartition", "--showfwinfo"],
                             capture_output=True, text=True, timeout=30).stdout
        info["rocm_smi"] = [ln.strip() for ln in out.splitlines() if ln.startswith("GPU[0]") and any(
            k in ln for k in ("Unique ID", "Partition", "MEC firmware", "RLC firmware", "SMC firmware", "SDMA firmware"))]
    except (OSError, subprocess.SubprocessError) as e:
        info["rocm_smi_error"] = str(e)
    info["amdgpu_parameters"] = {}
    for name in ("mtype_local", "enforce_isolation", "sched_policy", "hws_max_conc_proc", "mes", "cwsr_enable", "noretry", "mcbp"):
        try:     # how the host's driver caches local memory across the XCDs' L2s, isolates tenants, schedules queues
            with open(f"/sys/module/amdgpu/parameters/{name}") as f:
                info["amdgpu_parameters"][name] = f.read().strip()
        except OSError:
            pass
    os.makedirs(os.path.join(ROOT, "gpurun_out", "parity"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "parity", "box_info.json"), "w") as f:
        json.dump(info, f, indent=1)


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    _record_box()
    return torch.device("cuda:0")


AGREEMENT_DIR = os.path.join(ROOT, "gpurun_out", "parity")


def record_agreement(name, payload):
    """Write one measured-agreement record (robust / fragile counts, flips with their margins, max |dR| in 5-decimal
    units ...) to gpurun_out/parity/<name>.json on the GPU box; tools/collect_parity.py merges the records into the
    committed tests/golden/agreement_r06.json (earlier rounds: agreement_r02 / r03 / r05.json)."""
    import json
    os.makedirs(AGREEMENT_DIR, exist_ok=True)
    clean = {k: v for k, v in payload.items() if k != "same_mask"}
    with open(os.path.join(AGREEMENT_DIR, name.replace("/", "_") + ".json"), "w") as f:
        json.dump(clean, f, indent=1)


def eager_reference(pipe, svc, batch, **kw):
    """The single-stream eager pass a pipelined / graph-replayed result is compared with.  Its cooperative launches use the
    device's default workspaces, whose status nothing else in a test reads: it is CHECKED here (ops.check_status: failure codes
    and the proof of work, finished == expected workgroup-tiles) and a failed launch fails the test — no repeat (rounds 3-4
    repeated once with the write-through hand-off; a reference that needs a second attempt is a finding, not a warning)."""
    from gnnpn_sc_amd import ops
    out = pipe.run(svc, batch, **kw)
    ops.check_status(batch.x.device)
    return out
