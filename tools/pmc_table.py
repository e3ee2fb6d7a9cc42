"""Per-kernel table of DESIGN.md section 3.2 from a profile round (tools/profile_round.sh): GPU busy time, wait share, HBM bytes,
FP64 issued, LDS conflict share -- from pmc_summary.csv, pmc_traffic.json and kernel_stats_depth1_merge1.csv."""
import collections, csv, json, sys
d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_round"
acc = collections.defaultdict(dict)
for r in list(csv.reader(open(f"{d}/pmc_summary.csv")))[1:]:
    acc[r[0]][r[1]] = float(r[2])
tr = json.load(open(f"{d}/pmc_traffic.json"))
NB = 3          # batches in the depth-1 stats run of tools/profile_round.sh (--steps 2 --warmup 1)
ks = {r[0].split("(")[0]: float(r[2]) / 1e6 / NB for r in list(csv.reader(open(f"{d}/kernel_stats_depth1_merge1.csv")))[1:]}
tot_flop = 0
print("kernel            busy_ms  trace_ms  wait  HBM_GB  GFLOP  lds_conf")
for k in ("bmpc_k_ric", "bmpc_k_ric_att_thr", "bmpc_k_eval_main", "bmpc_k_eval_chain", "bmpc_k_eval", "bmpc_k_pose", "bmpc_k_step", "bmpc_k_trial", "bmpc_k_trial_spec", "bmpc_k_points", "bmpc_k_curv", "bmpc_k_fwd"):
    a = acc[k]
    busy = a.get("GRBM_GUI_ACTIVE", 0) / 8 / 2.4e9 * 1e3
    wait = a.get("SQ_WAIT_ANY", 0) / max(1, a.get("SQ_WAVE_CYCLES", 1))
    flop = (2 * a.get("SQ_INSTS_VALU_FMA_F64", 0) + a.get("SQ_INSTS_VALU_MUL_F64", 0) + a.get("SQ_INSTS_VALU_ADD_F64", 0)) * 64 / 1e9
    bc = a.get("SQ_LDS_BANK_CONFLICT", 0) / max(1, a.get("SQ_LDS_IDX_ACTIVE", 1))
    t = tr["per_kernel_GB"].get(k, {"fetch_raw": 0, "write": 0})
    print(f"{k:16s} {busy:8.1f} {ks.get(k, 0):9.1f} {wait:5.2f} {2 * t['fetch_raw'] + t['write']:7.1f} {flop:6.0f} {bc:8.2f}")
for a in acc.values():
    tot_flop += (2 * a.get("SQ_INSTS_VALU_FMA_F64", 0) + a.get("SQ_INSTS_VALU_MUL_F64", 0) + a.get("SQ_INSTS_VALU_ADD_F64", 0)) * 64 / 1e9
print(f"total FP64 issued {tot_flop:.0f} GFLOP; fetch raw {tr['fetch_bytes_raw'] / 1e9:.1f} GB, write {tr['write_bytes'] / 1e9:.1f} GB, "
      f"2*fetch+write {(2 * tr['fetch_bytes_raw'] + tr['write_bytes']) / 1e9:.1f} GB; sum of depth-1 kernel time per batch {sum(v for k, v in ks.items() if k.startswith('bmpc_')):.0f} ms")
