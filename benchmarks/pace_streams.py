"""
What two paced launches in flight on two streams cost each other: every paced launch takes its time
base from its own cell of g_pace_cells now (round 3: one global word for all kernels on all streams), so a launch's schedule
cannot be moved by another's read phase.  Measured here, cache-neutral (inputs rotating over sets larger than the Infinity
Cache), HIP events on each stream:

  alone       each kernel on its own: ms per launch, back to back
  pair        A on stream 1 and B on stream 2 at once, launch counts chosen so that both streams are busy about equally long;
              `makespan_over_sum` = time until both are done / (n_A x alone_A + n_B x alone_B): 1.0 = the pair costs exactly what
              the two cost one after the other (HBM-bound kernels cannot do better), above 1 = they disturb each other
  RK_PACE_SERIAL=0    the same with the library's cross-stream turn-taking of paced launches switched off (what two paced launches
              cost each other when they really overlap; the default makes a paced launch wait for the previous one on another stream)
  beside an exchange   the fan-out while another stream runs rk_comm_all_to_all (RCCL, world 1: the block is copied device to
              device) of configs[4]'s block size back to back

Prints one JSON object.
"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from librubiks_amd import _ffi, cube  # noqa: E402


def main():
	_ffi.check(_ffi.lib().rk_init(0))
	g = torch.Generator(device="cuda")
	g.manual_seed(2)
	N = 1_000_000
	par = [cube.device.apply_sequences(torch.randint(0, 12, (14, N), device="cuda", dtype=torch.uint8, generator=g), False, True) for _ in range(16)]   # 320 MB
	ch = [(torch.empty((12 * N, 20), dtype=torch.int8, device="cuda"), torch.empty(12 * N, dtype=torch.uint8, device="cuda")) for _ in range(2)]
	oh = [torch.empty((N, 480), dtype=torch.bfloat16, device="cuda") for _ in range(2)]
	M = 200_000
	cube.set_is2024(False)
	p6 = []
	solved = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(cube.get_solved(), (M, 6, 8, 6)))).cuda()
	for _ in range(6):
		p = solved
		for _ in range(4):
			p = cube.device.multi_rotate(p, torch.randint(0, 12, (M,), device="cuda", dtype=torch.uint8, generator=g))
		p6.append(p)
	ch6 = [(torch.empty((12 * M, 6, 8, 6), dtype=torch.int8, device="cuda"), torch.empty(12 * M, dtype=torch.uint8, device="cuda")) for _ in range(2)]
	cube.set_is2024(True)
	turn = {"fan": 0, "oh": 0, "f686": 0}

	def fan():
		i = turn["fan"]; turn["fan"] += 1
		cube.device.expand12(par[i % 16], *ch[i % 2])

	def onehot():
		i = turn["oh"]; turn["oh"] += 1
		cube.device.as_oh(par[(i + 5) % 16], oh[i % 2], torch.bfloat16)

	def fan686():
		i = turn["f686"]; turn["f686"] += 1
		_ffi.check(_ffi.lib().rk_expand12(_ffi.REPR_686, p6[i % 6].data_ptr(), ch6[i % 2][0].data_ptr(), ch6[i % 2][1].data_ptr(), None, M, _ffi.stream_ptr()))

	def alone(fn, n=40):
		for _ in range(5):
			fn()
		e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
		torch.cuda.synchronize()
		e0.record()
		for _ in range(n):
			fn()
		e1.record()
		torch.cuda.synchronize()
		return e0.elapsed_time(e1) / n

	s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

	def pair(fa, ta, fb, tb, total_ms=6.0):
		na, nb = max(4, int(total_ms / ta)), max(4, int(total_ms / tb))
		best = None
		for _ in range(5):
			torch.cuda.synchronize()
			e0, ea, eb = (torch.cuda.Event(enable_timing=True) for _ in range(3))
			e0.record()
			s1.wait_event(e0); s2.wait_event(e0)
			with torch.cuda.stream(s1):
				for _ in range(na):
					fa()
				ea.record()
			with torch.cuda.stream(s2):
				for _ in range(nb):
					fb()
				eb.record()
			torch.cuda.synchronize()
			span = max(e0.elapsed_time(ea), e0.elapsed_time(eb))
			best = span if best is None else min(best, span)
		return {"launches": [na, nb], "makespan_ms": best, "sum_alone_ms": na * ta + nb * tb, "makespan_over_sum": best / (na * ta + nb * tb)}

	out = {"bench": "pace_streams", "commit": os.environ.get("RK_COMMIT", ""),
	       "paced_launches_take_turns_across_streams": os.environ.get("RK_PACE_SERIAL", "1") != "0"}
	t_fan, t_oh, t_686 = alone(fan), alone(onehot), alone(fan686)
	out["alone_ms"] = {"fanout_1M": t_fan, "as_oh_bf16_1M": t_oh, "fanout686_200k": t_686}
	out["alone_frac_of_8TBps"] = {"fanout_1M": 272e6 / (t_fan * 1e-3) / 8e12, "as_oh_bf16_1M": 980e6 / (t_oh * 1e-3) / 8e12,
	                             "fanout686_200k": (288 + 3468) * M / (t_686 * 1e-3) / 8e12}
	out["fanout_with_as_oh"] = pair(fan, t_fan, onehot, t_oh)
	out["fanout_with_fanout686"] = pair(fan, t_fan, fan686, t_686)
	out["fanout_with_fanout"] = pair(fan, t_fan, fan, t_fan)
	_ffi.check(_ffi.lib().rk_set_pacing(0))
	u_fan, u_oh, u_686 = alone(fan), alone(onehot), alone(fan686)
	out["unpaced_alone_ms"] = {"fanout_1M": u_fan, "as_oh_bf16_1M": u_oh, "fanout686_200k": u_686}
	out["unpaced_fanout_with_as_oh"] = pair(fan, u_fan, onehot, u_oh)
	out["unpaced_fanout_with_fanout686"] = pair(fan, u_fan, fan686, u_686)
	_ffi.check(_ffi.lib().rk_set_pacing(-1))
	# beside an exchange: rk_comm_all_to_all at world 1 (RCCL really runs: the block is sent to and received from rank 0)
	try:
		from librubiks_amd.solving.sharded import RcclTransport
		tp = RcclTransport(RcclTransport.unique_id(), 0, 1)
		block = 32 + 8400 * 48                                               # configs[4]: N = 700
		snd, rcv = torch.zeros((1, block), dtype=torch.uint8, device="cuda"), torch.zeros((1, block), dtype=torch.uint8, device="cuda")
		def exchange():
			tp.all_to_all(snd, rcv)
		t_x = alone(exchange, 100)
		out["all_to_all_world1_block_bytes"] = block
		out["alone_ms"]["rk_comm_all_to_all_world1"] = t_x
		out["fanout_beside_all_to_all"] = pair(fan, t_fan, exchange, t_x)
		out["fanout_beside_all_to_all"]["fanout_ms_per_launch_if_exchange_were_free"] = t_fan
	except Exception as e:                                                   # RCCL unavailable: say so, keep the rest
		out["fanout_beside_all_to_all"] = {"error": repr(e)[:300]}
	print(json.dumps(out), flush=True)


if __name__ == "__main__":
	main()
