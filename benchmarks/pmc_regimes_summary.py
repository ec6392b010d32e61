"""
Condenses the rocprofv3 passes of benchmarks/pmc_regimes.py into one record per phase:

    python benchmarks/pmc_regimes_summary.py --manifest gpurun_out/x/manifest.json --passes gpurun_out/x/pmc_* --out profiles/r03_regimes_pmc.json

Every pass directory holds one `*_counter_collection.csv` (one row per dispatch and counter).  The fan-out kernel's dispatches
are taken in dispatch order and sliced by the manifest (warm-up launches of a phase are dropped); every counter is averaged
per launch.  Derived: average fabric read / write latency in TCC cycles = *_LEVEL / request count (the LEVEL counters
integrate the number of requests in flight over time).
"""
import argparse
import csv
import glob
import json
import os
import statistics

csv.field_size_limit(1 << 30)


def main():
	ap = argparse.ArgumentParser()
	ap.add_argument("--manifest", required=True)
	ap.add_argument("--passes", nargs="+", required=True)
	ap.add_argument("--kernel", default="k_expand12")
	ap.add_argument("--out")
	args = ap.parse_args()
	manifest = json.load(open(args.manifest))
	phases = {m["phase"]: dict(m, counters={}) for m in manifest}
	for d in args.passes:
		hits = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))
		if not hits:
			print(f"# no counter_collection.csv under {d} (pass failed?)")
			continue
		per_counter = {}
		with open(hits[-1], newline="") as f:
			for row in csv.DictReader(f):
				if args.kernel in row["Kernel_Name"] and "soa" not in row["Kernel_Name"]:
					per_counter.setdefault(row["Counter_Name"], []).append(
						(int(row["Dispatch_Id"]), float(row["Counter_Value"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
		for name, rows in per_counter.items():
			rows.sort()
			pos = 0
			for m in manifest:
				chunk = rows[pos + m["warm"]: pos + m["warm"] + m["launches"]]
				pos += m["warm"] + m["launches"]
				if chunk:
					phases[m["phase"]]["counters"][name] = statistics.fmean(v for _, v, _ in chunk)
					phases[m["phase"]].setdefault("dispatch_us_under_pmc", {})[name] = statistics.fmean(t for _, _, t in chunk) / 1e3
			if pos != len(rows):
				print(f"# {name}: {len(rows)} dispatches, manifest expects {pos}")
	for p in phases.values():
		c = p["counters"]
		d = {}
		if "TCC_EA0_RDREQ_sum" in c and "TCC_EA0_RDREQ_LEVEL_sum" in c and c["TCC_EA0_RDREQ_sum"]:
			d["avg_fabric_read_latency_tcc_cycles"] = c["TCC_EA0_RDREQ_LEVEL_sum"] / c["TCC_EA0_RDREQ_sum"]
		if "TCC_EA0_WRREQ_sum" in c and "TCC_EA0_WRREQ_LEVEL_sum" in c and c["TCC_EA0_WRREQ_sum"]:
			d["avg_fabric_write_latency_tcc_cycles"] = c["TCC_EA0_WRREQ_LEVEL_sum"] / c["TCC_EA0_WRREQ_sum"]
		if "TCP_UTCL1_TRANSLATION_MISS_sum" in c and c.get("TCP_UTCL1_REQUEST_sum"):
			d["utcl1_miss_rate"] = c["TCP_UTCL1_TRANSLATION_MISS_sum"] / c["TCP_UTCL1_REQUEST_sum"]
		if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c and (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]):
			d["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
		if c.get("TCP_TCC_WRITE_REQ_sum") and "TCP_TCC_WRITE_REQ_LATENCY_sum" in c:
			d["avg_tcp_write_latency_cycles"] = c["TCP_TCC_WRITE_REQ_LATENCY_sum"] / c["TCP_TCC_WRITE_REQ_sum"]
		if c.get("TCP_TCC_READ_REQ_sum") and "TCP_TCC_READ_REQ_LATENCY_sum" in c:
			d["avg_tcp_read_latency_cycles"] = c["TCP_TCC_READ_REQ_LATENCY_sum"] / c["TCP_TCC_READ_REQ_sum"]
		p["derived"] = d
		print(json.dumps(p))
	if args.out:
		with open(args.out, "w") as f:
			json.dump(list(phases.values()), f, indent=1)


if __name__ == "__main__":
	main()
