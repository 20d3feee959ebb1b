"""Condense rocprofv3 output (gpurun_out/prof_*) into the small tracked files under profiles/.

  python profiles/make_summary.py <round-tag> <trace dir> [<pmc dir>]

  profiles/<tag>_kernel_stats.csv     rocprofv3 --kernel-trace --stats summary (verbatim)
  profiles/<tag>_pmc.json             per-kernel averages of the PMC passes (FETCH_SIZE / WRITE_SIZE / MFMA counters)
  profiles/hbm_traffic.json           HBM bytes per k_linearize launch, read by bench.py (`roofline.traffic`)

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md section HBM: FETCH_SIZE and WRITE_SIZE are collected in SEPARATE
passes (TCC slots), both are reported in KiB, and on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide
(16 B / lane) coalesced stream, i.e. reads are under-reported 2x -> bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024.
"""
import collections
import csv
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def pmc_averages(path):
  acc = collections.defaultdict(lambda: collections.defaultdict(list))
  with open(path) as f:
    for r in csv.DictReader(f):
      name = r['Kernel_Name'].split('(')[0].replace('void ', '')
      acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
  return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


def first(pattern):
  import glob
  hits = sorted(glob.glob(pattern, recursive=True))
  return hits[0] if hits else None


def timeline(trace_csv, out_path, cfg):
  """consecutive kernels of two LM iterations from the kernel trace of profiles/scripts/prof_cfg.py (start / end relative to the first)"""
  rows = []
  with open(trace_csv) as f:
    for r in csv.DictReader(f):
      rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mcba::", ""),
                   r.get("Queue_Id", "?")))
  rows.sort()
  chol = [i for i, r in enumerate(rows) if r[2].startswith("k_chol_blk") or r[2].startswith("k_cholp_back")]
  if len(chol) < 8:
    return
  i0 = chol[len(chol) // 2]                      # the middle of the long solve
  i1 = chol[min(len(chol) - 1, len(chol) // 2 + 2)]
  t0 = rows[i0][0]
  with open(out_path, "w") as out:
    out.write(f"rocprofv3 --kernel-trace of profiles/scripts/prof_cfg.py {cfg} (long solve): consecutive kernels of two LM iterations, MI355X, HEAD\n\n")
    for s, e, name, q in rows[i0:i1 + 1]:
      out.write("%-34s queue %s  start %8.1f end %8.1f us  (%5.1f)\n" % (name[:34], q, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
    out.write("\niteration period: %.1f us\n" % ((rows[i1][0] - rows[i0][0]) / 2e3))


def collect_round(tag, root):
  """everything collect_r04.sh left under `root` -> profiles/<tag>_*"""
  main(tag, os.path.dirname(first(os.path.join(root, "trace", "**", "trace_kernel_stats.csv"))),
       os.path.dirname(first(os.path.join(root, "pmc", "**", "*_counter_collection.csv"))))
  for cfg in ("cfg2", "cfg3", "cfg4"):
    st = first(os.path.join(root, f"solve_{cfg}", "**", "solve_kernel_stats.csv"))
    if st:
      shutil.copy(st, os.path.join(HERE, f"{tag}_solve_kernel_stats_{cfg}.csv"))
    tr = first(os.path.join(root, f"solve_{cfg}", "**", "solve_kernel_trace.csv"))
    if tr and cfg in ("cfg3", "cfg4"):
      timeline(tr, os.path.join(HERE, f"{tag}_lm_timeline{'' if cfg == 'cfg3' else '_' + cfg}.txt"), cfg)
  out = {}
  for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = first(os.path.join(root, f"spmc_{c}", "**", "*_counter_collection.csv"))
    if f:
      for k, d in pmc_averages(f).items():
        out.setdefault(k, {}).update(d)
  if out:
    json.dump(out, open(os.path.join(HERE, f"{tag}_solve_pmc.json"), "w"), indent=1, sort_keys=True)
  # round 5: the LSMR iteration of the default solver (kernel statistics of the three forms, PMC averages of its kernels)
  st = first(os.path.join(root, "lsmr_trace", "**", "lsmr_kernel_stats.csv"))
  if st:
    shutil.copy(st, os.path.join(HERE, f"{tag}_lsmr_kernel_stats.csv"))
  out = {}
  for c in ("FETCH_SIZE", "WRITE_SIZE", "valu"):
    f = first(os.path.join(root, f"lsmr_pmc_{c}", "**", "*_counter_collection.csv"))
    if f:
      for k, d in pmc_averages(f).items():
        if "lsmr" in k:
          out.setdefault(k, {}).update(d)
  if out:
    json.dump(out, open(os.path.join(HERE, f"{tag}_lsmr_pmc.json"), "w"), indent=1, sort_keys=True)
  for src, dst in (("lsmr_iteration.log", f"{tag}_lsmr_iteration.txt"), ("bench_cfg4.json", f"{tag}_bench_cfg4.json"),
                   ("workspace_lsmr_cfg3.log", f"{tag}_workspace_calibrate_lsmr_cfg3.txt"),
                   ("workspace_lsmr_cfg4.log", f"{tag}_workspace_calibrate_lsmr_cfg4.txt"),
                   ("workspace_lsmr_cfg2.log", f"{tag}_workspace_calibrate_lsmr_cfg2.txt")):
    p = os.path.join(root, src)
    if os.path.exists(p) and os.path.getsize(p) > 0:
      shutil.copy(p, os.path.join(HERE, dst))
  for src, dst in (("bench.json", f"{tag}_bench.json"), ("lin_phases.log", f"{tag}_linearize_phases.txt"), ("lin_cfgs.log", f"{tag}_linearize_configs.txt"),
                   ("frame_groups.log", f"{tag}_frame_groups.txt"), ("chol_phases.log", f"{tag}_cholesky_phases.txt"),
                   ("chol_paths.log", f"{tag}_cholesky_paths.txt"), ("init.log", f"{tag}_initialise_poses.txt"),
                   ("workspace_cfg3.log", f"{tag}_workspace_calibrate_cfg3.txt"), ("workspace_cfg4.log", f"{tag}_workspace_calibrate_cfg4.txt"),
                   ("workspace_cfg2.log", f"{tag}_workspace_calibrate_cfg2.txt"),
                   ("workspace_cfg3_f32.log", f"{tag}_workspace_calibrate_cfg3_float32.txt"), ("parity_table.md", "parity_table.md"),
                   ("parity_table.json", "parity_table.json"), ("lsmr_mode.md", f"{tag}_lsmr_mode.md")):
    p = os.path.join(root, src)
    if os.path.exists(p) and os.path.getsize(p) > 0:
      shutil.copy(p, os.path.join(HERE, dst))


def main(tag, trace_dir, pmc_dir=None):
  shutil.copy(os.path.join(trace_dir, "trace_kernel_stats.csv"), os.path.join(HERE, f"{tag}_kernel_stats.csv"))
  if pmc_dir is None:
    return
  out = {}
  for fn in sorted(os.listdir(pmc_dir)):
    if fn.endswith("_counter_collection.csv"):
      for k, d in pmc_averages(os.path.join(pmc_dir, fn)).items():
        out.setdefault(k, {}).update(d)
  json.dump(out, open(os.path.join(HERE, f"{tag}_pmc.json"), "w"), indent=1, sort_keys=True)
  lin = [k for k in out if "k_linearize" in k]
  if lin:
    d = out[lin[0]]
    fetch_kib, write_kib = d.get("FETCH_SIZE"), d.get("WRITE_SIZE")
    traffic = dict(kernel=lin[0], FETCH_SIZE_KiB=fetch_kib, WRITE_SIZE_KiB=write_kib,
                   read_bytes_corrected=2 * fetch_kib * 1024, write_bytes=write_kib * 1024,
                   k_linearize_bytes_per_launch=2 * fetch_kib * 1024 + write_kib * 1024,
                   correction="gfx950: FETCH_SIZE under-reports wide coalesced reads 2x (MI355X_MICROARCH.md, HBM)",
                   source=f"profiles/{tag}_pmc.json")
    res = [k for k in out if "k_residual" in k]
    if res and out[res[0]].get("FETCH_SIZE") is not None and out[res[0]].get("WRITE_SIZE") is not None:
      traffic["k_residual_kernel"] = res[0]
      traffic["k_residual_bytes_per_launch"] = 2 * out[res[0]]["FETCH_SIZE"] * 1024 + out[res[0]]["WRITE_SIZE"] * 1024
    json.dump(traffic, open(os.path.join(HERE, "hbm_traffic.json"), "w"), indent=1)


if __name__ == "__main__":
  if len(sys.argv) == 3 and os.path.isdir(os.path.join(sys.argv[2], "trace")):
    collect_round(sys.argv[1], sys.argv[2])      # python profiles/make_summary.py r04 gpurun_out/r4p
  else:
    main(*sys.argv[1:4])
