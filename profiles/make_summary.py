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
    json.dump(traffic, open(os.path.join(HERE, "hbm_traffic.json"), "w"), indent=1)


if __name__ == "__main__":
  main(*sys.argv[1:4])
