"""
A/B of the fan-out kernel's shapes on one MI355X (interleaved, same process).  Every variant's output is checked
against the shipping kernel's; outputs rotate over 6 buffer sets (1.5 GB) so that nothing written stays in the
256 MiB Infinity Cache between launches -- the fair test for cached (plain) vs streaming (non-temporal) stores.
Uses the tuning hook rkx_expand12_variant (not part of the public C ABI).

Findings of round 1 (profiles/r01_tune_expand*.json): plain stores win only when the same 252 MB output is rewritten
in place (Infinity-Cache hits, 41.6 us) and lose when outputs rotate (49.9 us); non-temporal stores with 64-parent
tiles run at 44.5 us either way; an atomic tile counter on a persistent grid is 2-5x slower.
"""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from librubiks_amd import _ffi, cube  # noqa: E402

N = int(os.environ.get("RK_TUNE_N", "1000000"))
TUNE_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "librubiks_hip_tune.so")   # python -m librubiks_amd.build --tune
_ffi.LIB_PATH = TUNE_LIB if os.path.exists(TUNE_LIB) else sys.exit("build the tuning library first: python -m librubiks_amd.build --tune")
_ffi._lib = None          # drop the shipped library that importing the package loaded
lib = _ffi.lib()
lib.rkx_expand12_variant.restype = C.c_int
lib.rkx_expand12_variant.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p]

NAMES = {0: "nt, tile256", 1: "plain, tile256", 3: "plain, tile64",
         16: "nt, tile64, 4 waves/WG (shipping)", 17: "nt, tile64, 2 waves/WG", 18: "nt, tile64, 8 waves/WG", 19: "nt, tile64, 1 wave/WG",
         24: "nt, tile64, 4 waves/WG, software-pipelined input", 26: "nt, tile256, 4 waves/WG, pipelined input",
         27: "plain, tile64, 4 waves/WG, pipelined input",
         40: "GEOMETRY ONLY: loads + nt stores from registers, no LDS, no work", 41: "GEOMETRY ONLY: loads + LDS staging round trip + nt stores",
         44: "GEOMETRY ONLY: nt stores, no parent loads", 45: "GEOMETRY ONLY: nt stores, no loads, no flag stream (pure 240 MB store stream)",
         46: "GEOMETRY ONLY: plain stores, no parent loads", 47: "GEOMETRY ONLY: plain stores, no loads, no flag stream",
         50: "PURE STORES 240 MB: 15 KiB contiguous per wave, nt", 51: "PURE STORES: 15 KiB contiguous per wave, plain",
         52: "PURE STORES: 4 KiB contiguous per wave, nt", 53: "PURE STORES: 4 KiB contiguous per wave, plain",
         54: "PURE STORES: 4 KiB per wave interleaved in the workgroup, plain", 55: "PURE STORES: 15 KiB per wave interleaved, nt",
         56: "PURE STORES: 1 KiB per wave, plain", 57: "PURE STORES: 1 KiB per wave, nt", 58: "PURE STORES: 60 KiB contiguous per wave, nt",
         59: "PURE STORES: 16 KiB per wave interleaved, plain",
         80: "PURE STORES, run-time shape (grid_blocks = stores per wave | lanes of the last store << 8 | start offset in 16 B << 16), 4 waves/WG",
         81: "PURE STORES, run-time shape, 1 wave/WG",
         83: "PURE STORES, one store per wave after a life (grid_blocks = sleep x 64 clocks | mode << 8 | waves per workgroup << 12; mode 0 all waves sleep, 1 wave 0 sleeps + barrier, 2 a global load first)",
         84: "PURE STORES, run-time shape, 4 waves/WG, s_waitcnt vmcnt(0) after every store",
         86: "PURE STORES with the waves per CU bounded by an LDS allocation (grid_blocks = stores per wave | LDS KiB per workgroup << 8 | waves per workgroup << 16)",
         87: "PURE STORES 15 KiB per wave, stores held until a scheduled moment (grid_blocks = hold in 10 ns | stagger in 0.1 ns per wave << 12 | LDS KiB << 20 | 4 waves/WG << 28)",
         88: "PURE STORES 15 KiB per wave on one schedule for the launch: wave w stores from t0 + lead + w x tau on (grid_blocks = lead in 10 ns | tau in 0.01 ns << 10 | LDS KiB << 20 | 4 waves/WG << 28)",
         89: "GEOMETRY ONLY (parents in, children + flags out, no work), 64 parents per wave, stores on one schedule for the launch (grid_blocks as 88)",
         96: "GEOMETRY ONLY: children + flags out (no parent loads), stores on one schedule (grid_blocks as 88)",
         97: "GEOMETRY ONLY: parents in + children out (no flags), stores on one schedule (grid_blocks as 88)",
         98: "GEOMETRY ONLY: parents in + children out, the stores NOT depending on the loads, stores on one schedule (grid_blocks as 88)",
         99: "GEOMETRY ONLY, three streams, READ PHASE FIRST (the waves that start with the launch read all parents once), then stores on one schedule (grid_blocks as 88)",
         300: "PACED form: read phase by the first workgroups, then one tile per wave stored on a schedule (grid_blocks = lead in 10 ns | tau in 0.01 ns << 10 | pull workgroups / 64 << 20)",
         310: "PURE READS of the 240 MB children buffer (rotating over the output sets), 5 KiB per wave, one-shot grid, loads released on a schedule (grid_blocks = tau in 0.01 ns | LDS KiB << 12 | 4 waves/WG << 20)",
         311: "PURE READS, 1 KiB per wave (grid_blocks as 310)", 312: "PURE READS, 16 KiB per wave (grid_blocks as 310)",
         85: "PURE STORES, one 4 KiB page per 4-wave workgroup, pages of every aligned block visited with an odd stride (grid_blocks = log2(block pages) | stride << 8)",
         90: "GEOMETRY ONLY: 16-wave workgroup per 64 parents, loads + barrier + ONE store per wave (15 children pieces + flags)",
         91: "GEOMETRY ONLY: 8-wave workgroup per 64 parents, two stores per wave", 92: "GEOMETRY ONLY: 4-wave workgroup per 64 parents, four stores per wave",
         93: "GEOMETRY ONLY: 16-wave workgroups, persistent, next parents requested ahead", 94: "GEOMETRY ONLY: 8-wave workgroups, persistent",
         95: "GEOMETRY ONLY: 4-wave workgroups, persistent",
         82: "PURE STORES, one 4 KiB page per 4-wave workgroup, page of workgroup i rotated inside its group (grid_blocks = rot | group << 8)",
         60: "PURE STORES: 15 KiB per wave, nt, one-shot, back to back", 61: "PURE STORES: 15 KiB per wave, nt, s_sleep(1) between stores",
         62: "PURE STORES: 15 KiB per wave, nt, s_sleep(2) between stores", 63: "PURE STORES: 15 KiB per wave, nt, s_sleep(4) between stores",
         72: "GEOMETRY ONLY: 16 parents per wave (3.75 KiB), loads + stores + flags", 73: "GEOMETRY ONLY: 32 parents per wave (7.5 KiB), loads + stores + flags",
         74: "GEOMETRY ONLY: 8 parents per wave (1.9 KiB), loads + stores + flags", 75: "GEOMETRY ONLY: 16 parents per wave, children stream alone",
         76: "GEOMETRY ONLY: 32 parents per wave, children stream alone", 77: "GEOMETRY ONLY: 8 parents per wave, children stream alone",
         70: "GEOMETRY ONLY: one 1 KiB store per wave (16-wave workgroup per 64 parents: 15 chunk waves + 1 flag wave), nt",
         71: "GEOMETRY ONLY: one 1 KiB store per wave, plain",
         42: "GEOMETRY ONLY: plain stores from registers", 43: "GEOMETRY ONLY: LDS round trip + plain stores",
         28: "nt, tile64, 4 waves/WG, half-round staging (16 waves/CU)", 29: "nt, tile64, 4 waves/WG, half-round staging, pipelined input",
         100: "RING form, depth 0 (load, wait, expand; unconditional accesses)", 101: "RING form, 1 tile of parents in flight",
         102: "RING form, 2 tiles in flight", 104: "RING form, 4 tiles in flight", 108: "RING form, 8 tiles in flight",
         121: "RING form, 1 tile in flight, non-temporal parent loads", 122: "RING form, 2 tiles in flight, non-temporal parent loads",
         124: "RING form, 4 tiles in flight, non-temporal parent loads", 128: "RING form, 8 tiles in flight, non-temporal parent loads",
         200: "READ PHASE THEN WRITE PHASE: touch <= 8 M parents (pure read stream into the Infinity Cache), then one-shot ring-0 expand of them",
         202: "READ PHASE THEN WRITE PHASE: touch <= 8 M parents, then persistent ring-2 expand of them",
         152: "RING form, depth 0, stores sc1", 153: "RING form, depth 0, stores sc0 sc1 (write-through)", 154: "RING form, depth 0, stores sc1 nt",
         162: "RING form, 2 tiles in flight, stores sc1", 163: "RING form, 2 tiles in flight, stores sc0 sc1 (write-through)", 164: "RING form, 2 tiles in flight, stores sc1 nt",
         170: "RING form, depth 0 + 64 PULL workgroups in front (read the input once into the Infinity Cache)", 172: "RING form, 2 tiles in flight + 64 PULL workgroups",
         174: "RING form, 2 tiles in flight + 32 PULL workgroups", 176: "RING form, 2 tiles in flight + 128 PULL workgroups",
         141: "RING form, 1 tile in flight, plain stores", 142: "RING form, 2 tiles in flight, plain stores", 144: "RING form, 4 tiles in flight, plain stores"}
# (dropped from the code after losing clearly, results kept in profiles/r01_tune_expand*.json: atomic tile counter on a
#  persistent grid, per-lane strided input loads)


def main(variants):
	_ffi.check(lib.rk_init(0))
	g = torch.Generator(device="cuda")
	g.manual_seed(1)
	acts = torch.randint(0, 12, (20, N), device="cuda", dtype=torch.uint8, generator=g)
	parents = cube.device.apply_sequences(acts, False, True)
	ref_c, ref_f = cube.device.expand12(parents)
	counter = torch.zeros(4, dtype=torch.int64, device="cuda")
	bufs = [(torch.empty_like(ref_c), torch.empty_like(ref_f)) for _ in range(6 if N <= 2_000_000 else 2)]
	# inputs rotate too.  Round 3: over MORE than twice the 256 MiB Infinity Cache of distinct parents (RK_TUNE_IN_MB, default
	# 640 MB), so that a parent line cannot be a cache hit whatever the stores do to the cache; set 0 is `parents` (checked
	# against the shipping kernel), the others are further walks.
	n_in = max(2, -(-int(os.environ.get("RK_TUNE_IN_MB", "640")) * 1_000_000 // (20 * N)))
	ins = [parents] + [cube.device.apply_sequences(torch.randint(0, 12, (20, N), device="cuda", dtype=torch.uint8, generator=g), False, True)
	                   for _ in range(n_in - 1)]

	def run(vg, i):
		v, gb = vg
		c, f = bufs[i % len(bufs)]
		_ffi.check(lib.rkx_expand12_variant(v, ins[i % len(ins)].data_ptr(), c.data_ptr(), f.data_ptr(), None, N, counter.data_ptr(), gb, _ffi.stream_ptr()))

	res = {}
	for v in variants:
		bufs[0][0].zero_(); bufs[0][1].fill_(9)
		run(v, 0)
		ok = bool(torch.equal(bufs[0][0], ref_c) and torch.equal(bufs[0][1], ref_f)) if (v[0] < 40 or v[0] >= 100) else "n/a (diagnostic)"
		res[v] = {"variant": NAMES[v[0]], "id": v[0], "grid_blocks": v[1] or "default", "parents": N, "input_sets": len(ins), "output_sets": len(bufs),
		          "correct": ok, "ms": []}
		if v[0] in (80, 81, 84):
			last = (v[1] >> 8) & 255 or 64
			res[v]["shape"] = {"stores_per_wave": v[1] & 255, "lanes_of_last_store": last, "bytes_per_wave": ((v[1] & 255) - 1) * 1024 + last * 16,
			                   "start_offset_bytes": ((v[1] >> 16) & 0x7fff) * 16}
	launches = 3 * len(ins) if N <= 2_000_000 else 12
	for rep in range(7):
		for v in variants:
			for i in range(6):
				run(v, i)
			e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
			e0.record()
			for i in range(launches):
				run(v, i)
			e1.record()
			torch.cuda.synchronize()
			res[v]["ms"].append(round(e0.elapsed_time(e1) / launches, 5))
	for v in variants:
		r = res[v]
		ms = sorted(r["ms"])[len(r["ms"]) // 2]
		nbytes = (240.0 if 50 <= v[0] < 70 or v[0] in (80, 81, 82, 83, 84, 85, 86, 87, 88, 310, 311, 312) else 272.0) * N          # the pure-store diagnostics write the children buffer only
		r.update(ms_median=ms, **{"GB/s": round(nbytes / (ms * 1e-3) / 1e9, 1), "frac_of_8TBs": round(nbytes / (ms * 1e-3) / 8e12, 4)})
		print(json.dumps(r), flush=True)


if __name__ == "__main__":
	# arguments: variant or variant:grid_blocks
	args = [a.split(":") for a in sys.argv[1:]] or [["16"], ["17"], ["18"], ["19"], ["24"], ["24", "2048"], ["28"], ["0"], ["1"], ["3"]]
	main([(int(a[0]), int(a[1]) if len(a) > 1 else 0) for a in args])
