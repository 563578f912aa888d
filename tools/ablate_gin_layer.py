"""Where the split GIN layer's time goes: timing-only builds (results wrong by design) with one part removed each — compiled here,
timed on the GPU box.
    python tools/ablate_gin_layer.py build       # gnnpn-sc_amd/build/ablate/libgnnpn_hip_gin<bits>.so
    python tools/ablate_gin_layer.py run
bits (csrc/gin_layer_split.hip): 1 no weight stream (every k-block reads the first record), 2 no matrix instructions, 4 no
aggregate gathers, 8 no piece split in the epilogues."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "gnnpn-sc_amd")
OUT = os.path.join(PKG, "build", "ablate")
VARIANTS = [0, 1, 2, 3, 4, 8, 13, 15]
EXTRA = {}            # name -> extra -D flags (tools/ablate_gin_layer.py build NAME=-DFOO=1,-DBAR=2 ...)


def build():
    import importlib.util
    spec = importlib.util.spec_from_file_location("gnnpn_build", os.path.join(PKG, "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    b.build()
    os.makedirs(OUT, exist_ok=True)
    objs = [os.path.join(PKG, "build", s.replace(".hip", ".o")) for s in b.SOURCES]
    todo = {v: [f"-DGNNPN_GIN_ABLATE={v}"] for v in VARIANTS} if not EXTRA else EXTRA
    for v, flags in todo.items():
        o = os.path.join(OUT, f"gin_layer_split_abl{v}.o")
        subprocess.run(["hipcc"] + b.FLAGS + flags + ["-c", os.path.join(b.CSRC, "gin_layer_split.hip"), "-o", o], check=True)
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
