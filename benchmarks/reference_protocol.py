"""
The reference's own micro-benchmark (ref:librubiks/analysis/benchmark.py:22-130), both representations: single `rotate`,
`multi_rotate` / `as_oh` / `multi_is_solved` on 10 000 states, single `as_oh` and `is_solved` -- timed with its protocol
(every call timed alone, samples above twice the mean dropped, mean +- 1.96 s / sqrt(n); :92-103) on three surfaces:

  dropin   librubiks_amd.cube with NumPy arrays in and out, as reference code calls it (host -> device -> host per call);
  device   librubiks_amd.cube.device with the states resident in HBM (a stream synchronisation closes every timed call);
  numpy    the oracle's NumPy restatement of the reference's algorithm on this host (one core) -- the CPU beside it.  (For the
           6x8x6 `multi_rotate` the oracle gathers where the reference loops in Python: the reference itself moves 0.12 M
           states/s there)

The reference runs 1e7 single calls and 1e3 batched calls; the single calls are cut to 20 000 here (a call costs tens of
microseconds on every surface: the mean does not need 1e7 of them).

    python benchmarks/reference_protocol.py > profiles/r04_reference_protocol.json
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from librubiks_amd import cube  # noqa: E402
from oracle import cube_oracle as orc  # noqa: E402

SINGLE, BATCHED, STATES = 20_000, 1_000, 10_000


def protocol(samples) -> dict:
	"""benchmark.py:92-103 (Profile.remove_outliers, utils/ticktock.py:38-44: samples above threshold * mean are dropped once)"""
	x = np.asarray(samples)
	keep = x[x <= 2 * x.mean()]
	return {"mean_us": float(keep.mean() * 1e6), "ci95_us": float(1.96 * keep.std() / np.sqrt(len(keep)) * 1e6), "std_us": float(keep.std() * 1e6),
	        "removed": int(len(x) - len(keep)), "n": int(len(keep))}


def timed(calls):
	out = []
	for call in calls:
		t0 = time.perf_counter()
		call()
		out.append(time.perf_counter() - t0)
	return out


def walk_states(n_sets: int, n_states: int, is2024: bool) -> np.ndarray:
	"""benchmark.py:11-20: set 0 solved, every next set one random move further"""
	rot = orc.multi_rotate if is2024 else orc.multi_rotate686
	s = np.repeat((orc.SOLVED if is2024 else orc.solved_686())[None], n_states, axis=0)
	sets = [s]
	for _ in range(n_sets - 1):
		s = rot(s, np.random.randint(0, 6, n_states), np.random.randint(0, 2, n_states))
		sets.append(s)
	return np.array(sets)


def run(is2024: bool) -> dict:
	np.random.seed(0)
	rep = "20x24" if is2024 else "6x8x6"
	BATCHED = 1_000 if is2024 else 300           # (300 sets of 10 000 6x8x6 states are 0.9 GB)
	cube.set_is2024(is2024)
	o_rotate, o_multi = (orc.rotate, orc.multi_rotate) if is2024 else (orc.rotate686, orc.multi_rotate686)
	o_solved = orc.multi_is_solved if is2024 else orc.multi_is_solved686
	o_oh = orc.as_oh if is2024 else orc.as_oh686
	sync = torch.cuda.synchronize
	res = {}
	# single rotate (benchmark.py:27-36): a chain of moves from the solved state
	faces, dirs = np.random.randint(0, 6, SINGLE), np.random.randint(0, 2, SINGLE)
	for surface in ("dropin", "numpy"):
		state = [cube.get_solved() if surface == "dropin" else (orc.SOLVED if is2024 else orc.solved_686()).copy()]
		fn = cube.rotate if surface == "dropin" else o_rotate
		def step(f, d, state=state, fn=fn):
			state[0] = fn(state[0], f, d)
		res[f"rotate, 1 state, {surface}"] = protocol(timed([lambda f=f, d=d: step(f, d) for f, d in zip(faces[:SINGLE if surface == "dropin" else 5_000], dirs)]))
	# multi_rotate on 10 000 states (benchmark.py:38-48)
	faces, dirs = np.random.randint(0, 6, (BATCHED, STATES)), np.random.randint(0, 2, (BATCHED, STATES))
	s = [cube.repeat_state(cube.get_solved(), STATES)]
	def step(f, d):
		s[0] = cube.multi_rotate(s[0], f, d)
	res["multi_rotate, 10 000 states, dropin"] = protocol(timed([lambda f=f, d=d: step(f, d) for f, d in zip(faces, dirs)]))
	ds = [torch.from_numpy(cube.repeat_state(cube.get_solved(), STATES)).cuda()]
	acts = torch.from_numpy((2 * faces + (1 - dirs)).astype(np.uint8)).cuda()
	def dstep(i):
		ds[0] = cube.device.multi_rotate(ds[0], acts[i])
		sync()
	res["multi_rotate, 10 000 states, device"] = protocol(timed([lambda i=i: dstep(i) for i in range(BATCHED)]))
	so = [np.repeat((orc.SOLVED if is2024 else orc.solved_686())[None], STATES, axis=0)]
	def ostep(f, d):
		so[0] = o_multi(so[0], f, d)
	res["multi_rotate, 10 000 states, numpy"] = protocol(timed([lambda f=f, d=d: ostep(f, d) for f, d in zip(faces[:200], dirs)]))
	assert (s[0] == ds[0].cpu().numpy()).all()
	# one-hot and goal test, single and batched (benchmark.py:50-90)
	singles = walk_states(SINGLE, 1, is2024)[:, 0]
	batches = walk_states(BATCHED, STATES, is2024)
	dbatches = torch.from_numpy(batches).cuda()
	res["as_oh, 1 state, dropin"] = protocol(timed([lambda x=x: (cube.as_oh(x), sync()) for x in singles]))
	res["as_oh, 1 state, numpy"] = protocol(timed([lambda x=x: torch.from_numpy(o_oh(x[None])) for x in singles[:5_000]]))
	res["as_oh, 10 000 states, dropin"] = protocol(timed([lambda x=x: (cube.as_oh(x), sync()) for x in batches]))
	res["as_oh, 10 000 states, device"] = protocol(timed([lambda x=x: (cube.device.as_oh(x), sync()) for x in dbatches]))
	res["as_oh, 10 000 states, numpy"] = protocol(timed([lambda x=x: torch.from_numpy(o_oh(x)) for x in batches[:100]]))
	res["is_solved, 1 state, dropin"] = protocol(timed([lambda x=x: cube.is_solved(x) for x in singles]))
	res["is_solved, 1 state, numpy"] = protocol(timed([lambda x=x: bool(o_solved(x[None])[0]) for x in singles]))
	res["multi_is_solved, 10 000 states, dropin"] = protocol(timed([lambda x=x: cube.multi_is_solved(x) for x in batches]))
	res["multi_is_solved, 10 000 states, device"] = protocol(timed([lambda x=x: (cube.device.multi_is_solved(x), sync()) for x in dbatches]))
	res["multi_is_solved, 10 000 states, numpy"] = protocol(timed([lambda x=x: o_solved(x) for x in batches]))
	for k, v in res.items():
		if "10 000" in k:
			v["ns_per_state"] = v["mean_us"] * 1e3 / STATES
	return {"representation": rep, "results": res}


if __name__ == "__main__":
	cube.store_repr()
	out = [run(True), run(False)]
	cube.restore_repr()
	print(json.dumps({"bench": "reference_protocol", "protocol": "ref:librubiks/analysis/benchmark.py:22-130 (single calls cut from 1e7 to 20 000; NumPy legs shorter still)",
	                  "host": os.uname().nodename, "runs": out}))
