"""Where the split GIN layer's time goes: timing-only builds (results wrong by design) with one part removed each — compiled here,
timed on the GPU box.
    python tools/ablate_gin_layer.py build       # gnnpn-sc_amd/build/ablate/libgnnpn_hip_gin<bits>.so
    python tools/ablate_gin_layer.py run
The switches are NOT in the product sources (round 5): tools/experiments/gin_layer_split_switches.patch adds them to a copy of csrc/.
bits: 1 no weight stream (every k-block reads the first record), 2 no matrix instructions, 4 no
aggregate gathers, 8 no piece split in the epilogues."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "gnnpn-sc_amd")
OUT = os.path.join(PKG, "build", "ablate")
VARIANTS = [0, 1, 2, 3, 4, 8, 13, 15]
EXTRA = {}            # name -> extra -D flags (tools/ablate_gin_layer.py build NAME=-DFOO=1,-DBAR=2 ...)


def patched_csrc(patch):
    """A copy of csrc/ (+ include/) with tools/experiments/<patch> applied: the product sources carry no experiment switch, the
    timing-only builds are compiled from this copy."""
    import shutil
    dst = os.path.join(OUT, "src_" + patch.replace(".patch", ""))
    shutil.rmtree(dst, ignore_errors=True)
    os.makedirs(os.path.join(dst, "gnnpn-sc_amd"), exist_ok=True)
    shutil.copytree(os.path.join(PKG, "csrc"), os.path.join(dst, "gnnpn-sc_amd", "csrc"))
    subprocess.run(["git", "apply", "--unsafe-paths", "--directory=" + dst, os.path.join(ROOT, "tools", "experiments", patch)], check=True, cwd=ROOT)
    return os.path.join(dst, "gnnpn-sc_amd", "csrc")


def build():
    import importlib.util
    spec = importlib.util.spec_from_file_location("gnnpn_build", os.path.join(PKG, "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    b.build()
    os.makedirs(OUT, exist_ok=True)
    objs = [os.path.join(PKG, "build", s.replace(".hip", ".o")) for s in b.SOURCES]
    todo = {v: [f"-DGNNPN_GIN_ABLATE={v}"] for v in VARIANTS} if not EXTRA else EXTRA
    csrc = patched_csrc("gin_layer_split_switches.patch")
    flags0 = [f for f in b.FLAGS if f != "-I" + b.CSRC] + ["-I" + csrc]
    for v, flags in todo.items():
        o = os.path.join(OUT, f"gin_layer_split_abl{v}.o")
        subprocess.run(["hipcc"] + flags0 + flags + ["-c", os.path.join(csrc, "gin_layer_split.hip"), "-o", o], check=True)
        rest = [x for x in objs if os.path.basename(x) != "gin_layer_split.o"]
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(OUT, f"libgnnpn_hip_gin{v}.so")] + rest + [o], check=True)
        print("built", v, flush=True)


def run():
    for v in (EXTRA or VARIANTS):
        env = dict(os.environ, GNNPN_LIB=os.path.join(OUT, f"libgnnpn_hip_gin{v}.so"), PYTHONPATH=ROOT)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_gin_layer.py"), "--forms", "split"], env=env, capture_output=True, text=True)
        print(json.dumps({"ablate": v, "out": r.stdout.strip()[-400:], **({"err": r.stderr[-300:]} if r.returncode else {})}), flush=True)


if __name__ == "__main__":
    for a in sys.argv[2:]:
        name, _, flags = a.partition("=")
        EXTRA[name] = [f for f in flags.split(",") if f]
    build() if len(sys.argv) > 1 and sys.argv[1] == "build" else run()
