"""Copy the summaries produced by tools/r0N_profiles.sh (gpurun_out/r0Nprof/) into profiles/ (committed, what the judge
reads) and build profiles/r0N_pmc_traffic.json (per-launch HBM bytes from the FETCH_SIZE / WRITE_SIZE passes).
    python tools/collect_profiles.py [r04]"""
import collections, csv, glob, json, os, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RN = sys.argv[1] if len(sys.argv) > 1 else "r04"
O, P = os.path.join(R, "gpurun_out", RN + "prof"), os.path.join(R, "profiles")


def last_json_line(path):
    with open(path) as f:
        lines = [ln for ln in f.read().splitlines() if ln.startswith("{")]
    return json.loads(lines[-1]) if lines else None


def cp_json(src, dst):
    p = os.path.join(O, src)
    if os.path.exists(p) and os.path.getsize(p) > 0:
        d = last_json_line(p)
        if d is not None:
            json.dump(d, open(os.path.join(P, dst), "w"), indent=1)
            print("->", dst, d.get("value"))


for src, dst in (("bench_qws.json", RN + "_qws_b256_bench.json"), ("bench_qws_driver_flags.json", RN + "_qws_b256_bench_steps20.json"),
                 ("bench_normal.json", RN + "_normal_b1024_bench.json"), ("bench_synth4.json", RN + "_synth4_b512_bench.json"),
                 ("bench_synth4_strong_g4096_n1.json", RN + "_synth4_strong_g4096_n1_bench.json"),
                 ("bench_synth5.json", RN + "_synth5_b256_bench.json"), ("bench_synth5_f16.json", RN + "_synth5_b256_fp16_encoder_bench.json"),
                 ("bench_qws_f16.json", RN + "_qws_b256_fp16_encoder_bench.json"),
                 ("bench_force_dist_rccl_world1.json", RN + "_qws_b256_rccl_world1_bench.json"),
                 ("bench_selflaunch_4ranks_shared_gpu.json", RN + "_selflaunch_4ranks_shared_gpu_gloo_NOT_A_MEASUREMENT.json")):
    cp_json(src, dst)
if os.path.exists(os.path.join(O, "aggregate.jsonl")):
    recs = [json.loads(ln) for ln in open(os.path.join(O, "aggregate.jsonl")) if ln.startswith("{")]
    json.dump({"what": "GCN aggregate layer over B block-diagonal copies of the service graph (the reference's batching; edge lists in the reference's scan order), all forms, "
                       "tools/bench_aggregate.py, 1 x MI355X", "records": recs}, open(os.path.join(P, RN + "_csr_aggregate_roofline.json"), "w"), indent=1)
    print("->", RN + "_csr_aggregate_roofline.json")
stats = [(f"stats_{wl}_{pr}", f"{RN}_{key}_{pr}_solo_eager_kernel_stats.csv") for wl, key in (("qws", "qws_b256"), ("normal", "normal_b1024"),
                                                                                    ("synth4", "synth4_b512"), ("synth5", "synth5_b256"))
         for pr in ("split", "f32")]
stats += [("stats_qws_f16", RN + "_qws_b256_f16_solo_eager_kernel_stats.csv"), ("stats_synth5_f16", RN + "_synth5_b256_f16_solo_eager_kernel_stats.csv"),
          ("stats_qws_default", RN + "_qws_b256_default_cmd_kernel_stats.csv")]
for d, name in stats:
    f = sorted(glob.glob(f"{O}/{d}/*/*kernel_stats.csv"), key=os.path.getmtime)
    if f:
        shutil.copy(f[-1], os.path.join(P, name))
        print("->", name)
for src, dst in (("slot_parts_qws.txt", RN + "_slot_parts_qws_b256.txt"), ("stamps_decode.txt", RN + "_decode_phase_stamps.txt")):
    if os.path.exists(os.path.join(O, src)):
        shutil.copy(os.path.join(O, src), os.path.join(P, dst))
short = {"lstm_encode_coop_kernel": "lstm_encode", "pointer_decode_lean_kernel": "pointer_decode", "pointer_decode_coop_kernel": "pointer_decode", "gin_request_branch_kernel": "request_branch",
         "csr_aggregate_kernel<true>": "csr_aggregate_gcn", "select_candidates_kernel": "select_candidates", "select_candidates16_kernel": "select_candidates", "linear_f32_kernel<128": "linear_128",
         "linear_f32_kernel<64": "linear_64", "segment_mean_kernel": "segment_mean", "gin_layer_split_kernel<true>": "gin_layer1_split",
         "gin_layer_split_kernel<false>": "gin_layer0_split", "gin_layer_kernel<true>": "gin_layer1_f32", "gin_layer_kernel<false>": "gin_layer0_f32"}
sys.path.insert(0, R)
from gnnpn_sc_amd._lib import source_hash   # noqa: E402  (the tree the passes were measured on: collect BEFORE editing csrc/)
out = {"source_group": "recurrent", "source_hash": source_hash("recurrent"), "source_hash_all": source_hash(), "unit": "bytes per launch",
       "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and (separate pass) --pmc WRITE_SIZE over `bench.py --graph 0 --inflight 1` "
                 "(eager, one stream); counters are KB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced "
                 "reads); WRITE_SIZE as read; mean over the dispatches of the run"}
for wl, key0 in (("qws", "qws_b256"), ("normal", "normal_b1024"), ("synth4", "synth4_b512"), ("synth5", "synth5_b256")):
  for pr in ("split", "f32", "f16"):
    key = f"{key0}_{pr}"
    vals = collections.defaultdict(dict)
    for name in ("fetch", "write"):
        f = sorted(glob.glob(f"{O}/{name}_{wl}_{pr}/*/*counter_collection.csv"), key=os.path.getmtime)
        if not f:
            continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f[-1])):
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        with open(os.path.join(P, f"{RN}_{key}_pmc_{name}_size_summary.csv"), "w") as o:
            o.write("kernel,dispatches,mean_counter_value_KB\n")
            for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
                o.write(f"\"{k[:90]}\",{len(v)},{sum(v) / len(v):.1f}\n")
                vals[k][name] = sum(v) / len(v)
    if not vals:
        continue
    ks = {}
    for k, v in vals.items():
        for pat, label in short.items():
            if pat in k:
                fe, wr = v.get("fetch", 0) * 1024, v.get("write", 0) * 1024
                ks[label] = {"fetch_size_raw": round(fe), "fetch_corrected": round(2 * fe), "write_size": round(wr), "traffic": round(2 * fe + wr)}
    out[key] = {"workload": f"bench.py --workload {wl} --precision {pr} (default batch), eager single-stream launches, 1 x MI355X", "kernels": ks}
json.dump(out, open(os.path.join(P, RN + "_pmc_traffic.json"), "w"), indent=1)
print("->", RN + "_pmc_traffic.json", {k: list(v["kernels"]) for k, v in out.items() if isinstance(v, dict)})
