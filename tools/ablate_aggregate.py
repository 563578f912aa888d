"""Where the tiled aggregate's time goes: timing-only builds of the library (results wrong by design) with one part of the
kernel removed each — compiled here (no GPU needed), timed on the GPU box.

    python tools/ablate_aggregate.py build                 # compiles gnnpn-sc_amd/build/ablate/libgnnpn_hip_abl<bits>.so
    python tools/ablate_aggregate.py run [S:copies ...]    # on the GPU: times every build on the tiled form

The switches are NOT in the product sources (round 5): tools/experiments/aggregate_switches.patch adds them to a copy of csrc/.
bits: 1 no LDS reads, 2 one add instead of 2 multiplies + 2 adds per (edge, lane), 4 no tile fill,
8 no result stores, 16 no stream loads.
"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "gnnpn-sc_amd")
OUT = os.path.join(PKG, "build", "ablate")
VARIANTS = [0, 1, 2, 3, 4, 8, 12, 16, 19, 31]
EXTRA = {}            # name -> -D flags of an experiment build: python tools/ablate_aggregate.py build P1=-DGNNPN_TILED_PERSISTENT=1 ...


BASE = {"aggregate_switches.patch": "76d19b13d7", "aggregate_prefetch_wave.patch": "f5daec2"}   # the commit whose sources a frozen patch applies to (later kernels moved on)


def patched_csrc(patch):
    """A copy of csrc/ (+ include/) with tools/experiments/<patch> applied: the product sources carry no experiment switch, the
    timing-only builds are compiled from this copy.  A patch listed in BASE is frozen: the files it touches are taken from that
    commit (``git show``; needs the repository, i.e. build here, run on the GPU box)."""
    import re, shutil
    dst = os.path.join(OUT, "src_" + patch.replace(".patch", ""))
    shutil.rmtree(dst, ignore_errors=True)
    os.makedirs(os.path.join(dst, "gnnpn-sc_amd"), exist_ok=True)
    shutil.copytree(os.path.join(PKG, "csrc"), os.path.join(dst, "gnnpn-sc_amd", "csrc"))
    ppath = os.path.join(ROOT, "tools", "experiments", patch)
    if patch in BASE:
        for f in sorted(set(re.findall(r"^\+\+\+ b/(\S+)", open(ppath).read(), re.M))):
            blob = subprocess.run(["git", "show", f"{BASE[patch]}:{f}"], check=True, cwd=ROOT, capture_output=True).stdout
            os.makedirs(os.path.dirname(os.path.join(dst, f)), exist_ok=True)    # (a frozen patch may also touch the header, the oracle, a test)
            open(os.path.join(dst, f), "wb").write(blob)
    subprocess.run(["git", "apply", "--unsafe-paths", "--directory=" + dst, ppath], check=True, cwd=ROOT)
    return os.path.join(dst, "gnnpn-sc_amd", "csrc")


def build():
    sys.path.insert(0, PKG)
    import importlib.util
    spec = importlib.util.spec_from_file_location("gnnpn_build", os.path.join(PKG, "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    b.build()
    os.makedirs(OUT, exist_ok=True)
    objs = [os.path.join(PKG, "build", s.replace(".hip", ".o")) for s in b.SOURCES]
    todo = {v: [f"-DGNNPN_AGG_ABLATE={v}"] for v in VARIANTS} if not EXTRA else EXTRA
    csrc = patched_csrc("aggregate_switches.patch")
    flags0 = [f for f in b.FLAGS if f != "-I" + b.CSRC] + ["-I" + csrc]
    for v, flags in todo.items():
        mine = []
        for src in ("graph.hip", "graph_tiled.hip"):
            o = os.path.join(OUT, f"{src[:-4]}_abl{v}.o")
            subprocess.run(["hipcc"] + flags0 + flags + ["-c", os.path.join(csrc, src), "-o", o], check=True)
            mine.append(o)
        rest = [o for o in objs if os.path.basename(o) not in ("graph.o", "graph_tiled.o")]
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(OUT, f"libgnnpn_hip_abl{v}.so")] + rest + mine, check=True)
        print("built", v, flush=True)


def run(configs):
    for v in (EXTRA or VARIANTS):
        env = dict(os.environ, GNNPN_LIB=os.path.join(OUT, f"libgnnpn_hip_abl{v}.so"), PYTHONPATH=ROOT)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_aggregate.py"), "--configs", ",".join(configs),
                            "--forms", "tiled", "--no-check"], env=env, capture_output=True, text=True)
        for ln in r.stdout.splitlines():
            try:
                d = json.loads(ln)
            except ValueError:
                continue
            print(json.dumps({"ablate": v, "S": d["S"], "copies": d["copies"], "tiled_ms": d.get("tiled", {}).get("ms")}), flush=True)
        if r.returncode != 0:
            print(json.dumps({"ablate": v, "error": r.stderr[-400:]}), flush=True)


if __name__ == "__main__":
    rest = []
    for a in sys.argv[2:]:
        name, eq, flags = a.partition("=")
        if eq:
            EXTRA[name] = [f for f in flags.split(",") if f]
        else:
            rest.append(a)
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    else:
        run(rest or ["2507:256", "5000:128", "20000:8"])
