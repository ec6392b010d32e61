"""
GPU parity tests of the cube hot path: the HIP kernels (through the C ABI / the drop-in `cube` module) against
the committed golden vectors of the reference and against the CPU oracle on identical seeded inputs.
Bit-exact everywhere: all of this is integer / byte work (the one-hot is exact 0/1).
"""
import ctypes as C
import hashlib

import numpy as np
import pytest
import torch

from librubiks_amd import _ffi, cube
from oracle import c_oracle, cube_oracle as orc
from tests.helpers import random_walk, random_walk_c, sha

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _native_library_loaded():
	"""These tests are void if anything but the in-tree HIP library does the work."""
	lib = _ffi.lib()
	assert lib.rk_init(0) == 0, lib.rk_last_error()
	maps = open("/proc/self/maps").read()
	assert "librubiks_hip.so" in maps


def dev(a, dtype=None):
	t = torch.from_numpy(np.ascontiguousarray(a))
	if dtype is not None:
		t = t.to(dtype)
	return t.cuda()


# ------------------------------------------------------------------------------------------------- golden vectors
def test_golden_fanout_rotate_solved(golden):
	k = golden["cube_kat"]
	ch, fl = cube.expand(k["k3_256_parents"], return_solved=True)
	assert ch.dtype == np.int8 and ch.shape == (3072, 20)
	assert (ch == k["k3_256_children"]).all() and not fl.any()
	# exactly the reference idiom through the drop-in surface (agents.py:277-281)
	idiom = cube.multi_rotate(np.repeat(k["k3_256_parents"], 12, axis=0), *cube.iter_actions(256))
	assert (idiom == k["k3_256_children"]).all()
	out = cube.multi_rotate(k["k3_256_parents"], k["mr_faces"], k["mr_dirs"])
	assert out.dtype == np.int8 and (out == k["mr_out"]).all()
	assert (cube.multi_is_solved(k["solved_mix"]) == k["solved_mix_flags"]).all()
	_, fl = cube.expand(k["near_parents"], return_solved=True)
	assert (fl == k["near_children_solved"]).all()
	for i in range(0, 256, 31):
		assert (cube.rotate(k["k3_256_parents"][i], k["mr_faces"][i], k["mr_dirs"][i]) == k["mr_out"][i]).all()
	assert cube.is_solved(cube.get_solved()) and not cube.is_solved(k["k1_state"])


def test_golden_scramblers_and_onehot(golden):
	k = golden["cube_kat"]
	np.random.seed(0)
	s, f, d = cube.scramble(5)
	assert (s == k["k1_state"]).all() and (f == k["k1_faces"]).all() and (d == k["k1_dirs"]).all() and s.dtype == np.int8
	np.random.seed(42)
	s, f, d = cube.scramble(1)
	assert (s == k["k2a_state"]).all()
	s, f, d = cube.scramble(20)
	assert (s == k["k2b_state"]).all() and not cube.is_solved(s)
	for face, dr in zip(reversed(f), reversed([int(not x) for x in d])):     # tests/test_cube.py:112-114
		s = cube.rotate(s, face, dr)
	assert cube.is_solved(s)
	np.random.seed(7)
	s, f, d = cube.scramble(6, True)
	assert (s == k["k7_state"]).all()
	s, f, d = cube.scramble(0)
	assert cube.is_solved(s) and len(f) == 0

	np.random.seed(0)
	states, oh = cube.sequence_scrambler(4, 5, True)
	assert (states == k["k5_states"]).all() and oh.is_cuda and oh.dtype == torch.float32
	assert (oh.cpu().numpy() == k["k5_oh"]).all()
	np.random.seed(0)
	states, _ = cube.sequence_scrambler(3, 4, False)
	assert (states == k["k5b_states"]).all()
	one = cube.as_oh(k["k1_state"])
	assert one.shape == (1, 480) and (one.cpu().numpy() == k["oh_single"]).all()
	for dt in (torch.float16, torch.bfloat16):
		got = cube.device.as_oh(dev(k["k5_states"]), dtype=dt)
		assert (got.float().cpu().numpy() == k["k5_oh"]).all()


def test_reference_known_answer_test(golden):
	"""The body of the reference's tests/test_cube.py::_rotation_tests run against the drop-in, both representations."""
	text = golden["text"]
	for is2024 in (True, False):
		cube.set_is2024(is2024)
		state = cube.get_solved()
		assert cube.stringify(state) == text["str_solved"]
		for m, a in zip(((0, 1), (0, 0), (0, 1), (1, 1), (2, 0), (3, 0)), (False, True, False, False, False, False)):
			state = cube.rotate(state, *m)
			assert a == cube.is_solved(state)
		for m, a in zip(((3, 1), (2, 1), (1, 0), (0, 0)), (False, False, False, True)):
			state = cube.rotate(state, *m)
			assert a == cube.is_solved(state)
		assert cube.stringify(cube.rotate(cube.get_solved(), 0, 1)) == text["str_F"]
		state = cube.get_solved()
		for m in ((0, 0), (1, 0), (2, 0), (3, 0), (4, 0), (5, 0), (0, 1), (1, 1), (2, 1), (3, 1), (4, 1), (5, 1)):
			state = cube.rotate(state, *m)
			assert not cube.is_solved(state)
		assert cube.stringify(state) == text["str_all12"]
		# _multi_rotate_test (tests/test_cube.py:94-101), both directions
		np.random.seed(3)
		states = np.array([cube.get_solved()] * 5)
		for _ in range(10):
			faces, dirs = np.random.randint(0, 6, 5), np.random.randint(0, 2, 5)
			classic = np.array([cube.rotate(s, f, d) for s, f, d in zip(states, faces, dirs)])
			states = cube.multi_rotate(states, faces, dirs)
			assert (classic == states).all()


def test_repr686_golden(golden):
	k = golden["cube_kat"]
	cube.set_is2024(False)
	s6 = k["r686_states"]
	assert (cube.multi_rotate(s6, k["r686_faces"], k["r686_dirs"]) == k["r686_out"]).all()
	ch, fl = cube.expand(s6[:8], return_solved=True)
	assert (ch.reshape(8, 12, 6, 8, 6) == k["r686_all12"]).all() and not fl.any()
	assert (cube.as_correct(torch.from_numpy(s6)).cpu().numpy() == k["r686_correct"]).all()
	st = cube.rotate(cube.rotate(cube.get_solved(), 0, True), 5, False)
	assert (cube.as_correct(torch.from_numpy(st).unsqueeze(0)).cpu().numpy() == k["r686_correct_FRp"]).all()
	oh = cube.as_oh(s6)
	assert oh.shape == (64, 288) and (oh.cpu().numpy().reshape(s6.shape) == s6).all()
	assert cube.as_oh(s6[0]).shape == (1, 288)
	near = cube.multi_rotate(cube.repeat_state(cube.get_solved(), 12), *cube.iter_actions())
	assert not cube.multi_is_solved(near).any()
	_, fl = cube.expand(near, return_solved=True)
	assert fl.sum() == 12 and all(fl[12 * a + (a ^ 1)] for a in range(12))
	np.random.seed(11)
	s, f, d = cube.scramble(9)
	ref = orc.SOLVED686
	for face, dr in zip(f, d):
		ref = orc.rotate686(ref, face, dr)
	assert (s == ref).all()


@pytest.mark.parametrize("n", [1, 3, 4, 5, 7, 257, 4099])
def test_repr686_fanout_flags_one_launch(n):
	"""6x8x6 fan-out with the goal test fused into the same launch: children, flags, count and first index against the
	oracle, with parents one move from solved, solved parents and slots that are NOT one-hot (must never read as solved)."""
	cube.set_is2024(False)
	np.random.seed(600 + n)
	p = np.broadcast_to(orc.SOLVED686, (n, 6, 8, 6)).copy()
	for _ in range(7):
		p = orc.multi_rotate686(p, np.random.randint(0, 6, n), np.random.randint(0, 2, n))
	p[n // 2] = orc.rotate686(orc.SOLVED686, 4, 0)                    # child `rev` of this parent is solved
	if n > 2:
		p[0] = orc.SOLVED686                                         # a solved parent has no solved child
		p[n - 1] = orc.rotate686(orc.SOLVED686, 2, 1)
	if n > 6:
		junk = orc.rotate686(orc.SOLVED686, 1, 1).copy()
		junk[3, 5] = [1, 0, 0, 1, 0, 0]                              # two ones in a slot
		p[1] = junk
		junk2 = orc.rotate686(orc.SOLVED686, 5, 0).copy()
		junk2[0, 0] = 0                                              # an empty slot
		junk2[0, 0, 2] = 2                                           # and a byte that is not 0/1
		p[2] = junk2
	ref_ch = np.stack([orc.rotate686(s, a // 2, 1 - a % 2) for s in p for a in range(12)])
	ref_fl = orc.multi_is_solved686(ref_ch)
	assert ref_fl.sum() >= (1 if n <= 2 else 2)                      # the planted ones (a random walk may add its own)
	stats = torch.tensor([0, _ffi.INT64_MAX], dtype=torch.int64, device="cuda")
	ch, fl = cube.device.expand12(dev(p), stats=stats)
	assert (ch.cpu().numpy() == ref_ch).all() and (fl.cpu().numpy().astype(bool) == ref_fl).all()
	st = stats.cpu().numpy()
	assert st[0] == ref_fl.sum() and st[1] == np.flatnonzero(ref_fl)[0]
	ch2, none = cube.device.expand12(dev(p), want_flags=False)
	assert none is None and (ch2.cpu().numpy() == ref_ch).all()


# ------------------------------------------------------------------------------------------------- oracle parity
@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 255, 256, 257, 1000, 4099, 70_001])
def test_ragged_sizes_against_oracle(n):
	"""Tile boundaries of every kernel: 64-parent rounds, 256-state tiles, partial tails."""
	p = random_walk(n, 12, seed=100 + n % 97)
	# sprinkle solved states and near-solved parents so that flags / stats paths fire
	if n > 3:
		p[n // 3] = orc.SOLVED
		p[-1] = orc.rotate(orc.SOLVED, 3, 1)
	dp = dev(p)
	stats = torch.tensor([0, _ffi.INT64_MAX], dtype=torch.int64, device="cuda")
	ch, fl = cube.device.expand12(dp, stats=stats)
	ref_ch, ref_fl = c_oracle.expand12(p, threads=4)
	assert (ch.cpu().numpy() == ref_ch).all()
	assert (fl.cpu().numpy() == ref_fl).all()
	st = stats.cpu().numpy()
	assert st[0] == ref_fl.sum()
	assert st[1] == (np.flatnonzero(ref_fl)[0] if ref_fl.any() else _ffi.INT64_MAX)
	ch2, none = cube.device.expand12(dp, want_flags=False)
	assert none is None and (ch2.cpu().numpy() == ref_ch).all()

	np.random.seed(n)
	acts = np.random.randint(0, 12, n).astype(np.uint8)
	out = cube.device.multi_rotate(dp, dev(acts))
	assert (out.cpu().numpy() == c_oracle.multi_rotate(p, acts)).all()
	assert (dp.cpu().numpy() == p).all()                                   # inputs are never mutated
	stats = torch.tensor([0, _ffi.INT64_MAX], dtype=torch.int64, device="cuda")
	flags = cube.device.multi_is_solved(dp, stats=stats)
	ref_flags, cnt, first = c_oracle.multi_is_solved(p)
	assert (flags.cpu().numpy().astype(bool) == ref_flags).all()
	st = stats.cpu().numpy()
	assert st[0] == cnt and st[1] == (first if cnt else _ffi.INT64_MAX)
	assert (cube.device.as_oh(dp).cpu().numpy() == c_oracle.as_oh(p)).all()


def test_misaligned_views_and_inplace():
	"""Row slices of a state pool are only 4-byte aligned: the kernels must take the dword path, not fault."""
	p = random_walk(3000, 9, seed=5)
	pool = dev(np.concatenate([np.zeros((1, 20), np.int8), p]))
	view = pool[1:]                                                        # base + 20 B
	assert view.data_ptr() % 16 != 0
	ref_ch, ref_fl = c_oracle.expand12(p)
	ch, fl = cube.device.expand12(view)
	assert (ch.cpu().numpy() == ref_ch).all() and (fl.cpu().numpy() == ref_fl).all()
	acts = (np.arange(3000) % 12).astype(np.uint8)
	ref = c_oracle.multi_rotate(p, acts)
	out_pool = torch.zeros_like(pool)
	cube.device.multi_rotate(view, dev(acts), out=out_pool[1:])
	assert (out_pool[1:].cpu().numpy() == ref).all() and (out_pool[0] == 0).all()
	# in place
	cube.device.multi_rotate(view, dev(acts), out=view)
	assert (view.cpu().numpy() == ref).all()
	assert (cube.device.multi_is_solved(view).cpu().numpy().astype(bool) == orc.multi_is_solved(ref)).all()
	assert (cube.device.as_oh(view).cpu().numpy() == orc.as_oh(ref)).all()
	# flags into a view that is 4- but not 16-byte aligned
	fl_pool = torch.zeros(12 * 3000 + 4, dtype=torch.uint8, device="cuda")
	near = dev(np.concatenate([orc.rotate(orc.SOLVED, 2, 0)[None], ref[:2999]]))
	_, fl = cube.device.expand12(near, solved=fl_pool[4:])
	want = c_oracle.expand12(near.cpu().numpy())[1]
	assert (fl.cpu().numpy() == want).all() and want.sum() >= 1 and (fl_pool[:4] == 0).all()


def test_empty_and_errors():
	assert cube.multi_rotate(np.zeros((0, 20), np.int8), [], []).shape == (0, 20)
	assert cube.multi_is_solved(np.zeros((0, 20), np.int8)).shape == (0,)
	with pytest.raises(IndexError):
		cube.multi_rotate(np.zeros((2, 20), np.int8), [0, 6], [0, 1])
	with pytest.raises(IndexError):
		cube.multi_rotate(np.zeros((2, 20), np.int8), [0, 1], [0, 2])
	lib = _ffi.lib()
	assert lib.rk_expand12(0, None, None, None, None, 0, None) == 0          # n = 0 is a no-op
	# the host-pointer C entries (what a ctypes-only caller would use)
	p = random_walk(777, 8, seed=9)
	ch = np.empty((12 * 777, 20), np.int8)
	fl = np.empty(12 * 777, np.uint8)
	st = np.zeros(2, np.int64)
	_ffi.check(lib.rk_expand12_host(0, p.ctypes.data, ch.ctypes.data, fl.ctypes.data, st.ctypes.data, 777, None))
	assert (ch == orc.expand12(p)).all() and not fl.any() and st.tolist() == [0, -1]
	acts = (np.arange(777) % 12).astype(np.uint8)
	out = np.empty_like(p)
	_ffi.check(lib.rk_multi_rotate_host(0, p.ctypes.data, acts.ctypes.data, out.ctypes.data, 777, None))
	assert (out == c_oracle.multi_rotate(p, acts)).all()
	acts[5] = 12
	assert lib.rk_multi_rotate_host(0, p.ctypes.data, acts.ctypes.data, out.ctypes.data, 777, None) == -1
	p[100] = orc.SOLVED
	_ffi.check(lib.rk_multi_is_solved_host(0, p.ctypes.data, fl.ctypes.data, st.ctypes.data, 777, None))
	assert fl[:777].sum() == 1 and st.tolist() == [1, 100]
	seq = np.random.randint(0, 12, (6, 9)).astype(np.uint8)
	got = np.empty((9 * 6, 20), np.int8)
	_ffi.check(lib.rk_apply_sequences_host(0, seq.ctypes.data, 6, 9, 0, 0, got.ctypes.data, None))
	assert (got == orc.sequence_states(seq // 2, 1 - seq % 2, False)).all()


@pytest.mark.parametrize("n", [1, 777, 26_000, 70_001])
def test_host_entries_zero_copy_and_staged(n):
	"""The rk_*_host entries on both sides of the zero-copy limit (1 MiB of arrays per call: below it the arrays pass through a
	page-locked, device-mapped buffer, above it through device scratch) -- same results; and rk_as_oh_host (host states in,
	device one-hot out) in both representations and output types."""
	lib = _ffi.lib()
	p = random_walk_c(n, 9, seed=n)
	p[n // 2] = orc.SOLVED
	acts = np.random.RandomState(n).randint(0, 12, n).astype(np.uint8)
	out, fl = np.empty_like(p), np.empty(n, np.uint8)
	_ffi.check(lib.rk_multi_rotate_host(0, p.ctypes.data, acts.ctypes.data, out.ctypes.data, n, None))
	assert (out == c_oracle.multi_rotate(p, acts)).all()
	_ffi.check(lib.rk_multi_is_solved_host(0, p.ctypes.data, fl.ctypes.data, None, n, None))
	assert (fl.astype(bool) == orc.multi_is_solved(p)).all()
	st = np.zeros(2, np.int64)
	_ffi.check(lib.rk_multi_is_solved_host(0, p.ctypes.data, fl.ctypes.data, st.ctypes.data, n, None))     # with the counters: staged
	assert st.tolist() == [int(orc.multi_is_solved(p).sum()), int(np.argmax(orc.multi_is_solved(p)))]
	m = min(n, 9_000)
	ch, cf = np.empty((12 * m, 20), np.int8), np.empty(12 * m, np.uint8)
	_ffi.check(lib.rk_expand12_host(0, p.ctypes.data, ch.ctypes.data, cf.ctypes.data, None, m, None))
	ref_ch, ref_fl = c_oracle.expand12(p[:m])
	assert (ch == ref_ch).all() and (cf == ref_fl).all()
	for dtype, code in ((torch.float32, _ffi.OH_F32), (torch.bfloat16, _ffi.OH_BF16)):
		oh = torch.empty((n, 480), dtype=dtype, device="cuda")
		_ffi.check(lib.rk_as_oh_host(0, p.ctypes.data, oh.data_ptr(), code, n, None))
		assert (oh.float().cpu().numpy() == c_oracle.as_oh(p)).all()
	assert (cube.as_oh(p).cpu().numpy() == c_oracle.as_oh(p)).all() and cube.as_oh(p[0]).shape == (1, 480)
	k = min(n, 5_000)
	s686 = orc.solved_686()[None].repeat(k, axis=0)
	s686 = c_oracle.multi_rotate686(s686, acts[:k])
	oh = torch.empty((k, 288), dtype=torch.float32, device="cuda")
	_ffi.check(lib.rk_as_oh_host(1, s686.ctypes.data, oh.data_ptr(), _ffi.OH_F32, k, None))
	assert (oh.cpu().numpy() == orc.as_oh686(s686)).all()
	assert lib.rk_as_oh_host(0, None, oh.data_ptr(), _ffi.OH_F32, 3, None) == -1 and lib.rk_as_oh_host(0, p.ctypes.data, oh.data_ptr(), 7, 3, None) == -1
	assert lib.rk_as_oh_host(0, None, None, _ffi.OH_F32, 0, None) == 0


def test_device_tensors_round_trip_through_dropin_surface():
	p = random_walk(500, 7, seed=21)
	dp = dev(p)
	f, d = np.random.randint(0, 6, 500), np.random.randint(0, 2, 500)
	out = cube.multi_rotate(dp, dev(f), dev(d))
	assert out.is_cuda and (out.cpu().numpy() == orc.multi_rotate(p, f, d)).all()
	assert cube.multi_is_solved(dp).dtype == torch.bool
	ch = cube.expand(dp)
	assert ch.is_cuda and (ch.cpu().numpy() == orc.expand12(p)).all()


# ------------------------------------------------------------------------------------------------- full size
def test_headline_1m_bit_exact(golden):
	"""
	BASELINE config 2: 1 M seeded scrambles x 12 moves, children bit-identical to the reference: SHA-256 of the
	240 MB children array equals the hash the reference produced (tests/golden/cube_text.json), plus a direct
	comparison with the C oracle and the size-independent properties.
	"""
	text = golden["text"]
	n = 1_000_000
	p = random_walk_c(n, 20, seed=1)
	assert sha(p) == text["k3_1m_parents_sha256"]
	dp = dev(p)
	stats = torch.tensor([0, _ffi.INT64_MAX], dtype=torch.int64, device="cuda")
	ch, fl = cube.device.expand12(dp, stats=stats)
	ch_h = ch.cpu().numpy()
	assert hashlib.sha256(ch_h.tobytes()).hexdigest() == text["k3_1m_children_sha256"]
	assert int(fl.sum()) == text["k3_1m_solved_children"] == 0 and stats[0].item() == 0
	ref, _ = c_oracle.expand12(p, threads=8)
	assert (ch_h == ref).all()
	del ref, ch_h

	# property: child a, moved back by rev(a), is the parent again -- for all 12 M children
	rev = torch.arange(12, dtype=torch.uint8, device="cuda").bitwise_xor(1).repeat(n)
	back = cube.device.multi_rotate(ch, rev)
	assert torch.equal(back.view(n, 12, 20), dp.view(n, 1, 20).expand(n, 12, 20))
	# property: applying the 12 moves per row with multi_rotate gives the same children (two different kernels)
	acts = torch.arange(12, dtype=torch.uint8, device="cuda").repeat(n)
	via_rows = cube.device.multi_rotate(dp.repeat_interleave(12, dim=0), acts)
	assert torch.equal(via_rows, ch)
	# property: a depth-20 walk undone move by move ends solved everywhere
	np.random.seed(77)
	acts_seq = np.random.randint(0, 12, (20, n)).astype(np.uint8)
	cur = dev(orc.repeat_state(orc.SOLVED, n))
	for dmove in range(20):
		cur = cube.device.multi_rotate(cur, dev(acts_seq[dmove]))
	assert int(cube.device.multi_is_solved(cur).sum()) < n // 1000
	walk = cube.device.apply_sequences(dev(acts_seq), with_solved=False, only_last=True)
	assert torch.equal(walk, cur)
	for dmove in reversed(range(20)):
		cur = cube.device.multi_rotate(cur, dev(acts_seq[dmove] ^ 1))
	stats = torch.tensor([0, _ffi.INT64_MAX], dtype=torch.int64, device="cuda")
	flags = cube.device.multi_is_solved(cur, stats=stats)
	assert int(flags.sum()) == n and stats.tolist() == [n, 0]
	# one-hot rows: exactly 20 ones, in the right columns (checked as an index round trip)
	oh = cube.device.as_oh(dp[:200_000])
	assert torch.equal(oh.sum(dim=1), torch.full((200_000,), 20.0, device="cuda"))
	cols = oh.view(200_000, 20, 24).argmax(dim=2).to(torch.int8)
	assert torch.equal(cols, dp[:200_000])


def test_fuzz_shapes_offsets_against_oracle():
	"""Random sizes and pool offsets (4-byte-aligned views) through every 20-byte kernel, device path."""
	rng = np.random.RandomState(2024)
	pool_rows = 9000
	base = random_walk(pool_rows, 11, seed=77)
	base[rng.randint(0, pool_rows, 40)] = orc.SOLVED
	near = orc.multi_rotate(orc.repeat_state(orc.SOLVED, 12), *orc.iter_actions())
	base[rng.randint(0, pool_rows, 60)] = near[rng.randint(0, 12, 60)]
	pool = dev(base)
	for _ in range(40):
		n = int(rng.choice([1, 3, 64, 65, 127, 128, 300, 511, 1024, 2049, 5000]))
		off = int(rng.randint(0, pool_rows - n))
		view, host = pool[off:off + n], base[off:off + n]
		ch, fl = cube.device.expand12(view)
		ref_ch, ref_fl = c_oracle.expand12(host)
		assert (ch.cpu().numpy() == ref_ch).all() and (fl.cpu().numpy() == ref_fl).all(), (n, off)
		acts = rng.randint(0, 12, n).astype(np.uint8)
		assert (cube.device.multi_rotate(view, dev(acts)).cpu().numpy() == c_oracle.multi_rotate(host, acts)).all(), (n, off)
		assert (cube.device.multi_is_solved(view).cpu().numpy().astype(bool) == orc.multi_is_solved(host)).all(), (n, off)
		assert (cube.device.as_oh(view).cpu().numpy() == c_oracle.as_oh(host)).all(), (n, off)
		depth, games = int(rng.randint(1, 9)), int(rng.randint(1, 70))
		seq = rng.randint(0, 12, (depth, games)).astype(np.uint8)
		for ws in (False, True):
			got = cube.device.apply_sequences(dev(seq), ws, False).cpu().numpy()
			assert (got == orc.sequence_states(seq // 2, 1 - seq % 2, ws)).all(), (depth, games, ws)


@pytest.mark.parametrize("n", [1, 63, 256, 1000, 70_001])
def test_soa_fanout_matches_aos(n):
	"""The structure-of-arrays form: same children and flags as the AoS kernel / the oracle, in plane layout."""
	p = random_walk(n, 10, seed=300 + n % 13)
	if n > 5:
		p[n // 2] = orc.rotate(orc.SOLVED, 5, 1)
		p[1] = orc.SOLVED
	dp = dev(p)
	planes = cube.device.to_soa(dp)
	assert planes.shape == (5, n)
	assert (planes.cpu().numpy().T.copy().view(np.int8).reshape(n, 20) == p).all()
	assert torch.equal(cube.device.from_soa(planes), dp)
	stats = torch.tensor([0, _ffi.INT64_MAX], dtype=torch.int64, device="cuda")
	ch, fl = cube.device.expand12_soa(planes, stats=stats)
	ref_ch, ref_fl = c_oracle.expand12(p)
	got = cube.device.from_soa(ch).cpu().numpy()                       # (12, n, 20): action-major
	assert (got.transpose(1, 0, 2).reshape(12 * n, 20) == ref_ch).all()
	assert (fl.cpu().numpy().T.reshape(-1) == ref_fl).all()
	st = stats.cpu().numpy()
	assert st[0] == ref_fl.sum() and st[1] == (np.flatnonzero(ref_fl)[0] if ref_fl.any() else _ffi.INT64_MAX)


def test_large_batch_persistent_path():
	"""Above 8.4 M parents the launcher switches to the persistent, input-prefetching grid: same results."""
	n = 8_500_000 + 37
	g = torch.Generator(device="cuda")
	g.manual_seed(5)
	acts = torch.randint(0, 12, (18, n), device="cuda", dtype=torch.uint8, generator=g)
	parents = cube.device.apply_sequences(acts, False, True)
	del acts
	parents[12345] = dev(orc.rotate(orc.SOLVED, 0, 1))
	parents[n - 1] = dev(orc.rotate(orc.SOLVED, 3, 0))
	stats = torch.tensor([0, _ffi.INT64_MAX], dtype=torch.int64, device="cuda")
	ch, fl = cube.device.expand12(parents, stats=stats)
	# a sample against the oracle, everything against the inverse-move property
	pick = torch.cat([torch.arange(0, 4096), torch.randint(0, n, (4096,)), torch.arange(n - 4096, n)]).cuda()
	ref_ch, ref_fl = c_oracle.expand12(parents[pick].cpu().numpy())
	got = ch.view(n, 12, 20)[pick].reshape(-1, 20).cpu().numpy()
	assert (got == ref_ch).all()
	assert (fl.view(n, 12)[pick].reshape(-1).cpu().numpy() == ref_fl).all()
	rev = torch.arange(12, dtype=torch.uint8, device="cuda").bitwise_xor(1).repeat(n)
	back = cube.device.multi_rotate(ch, rev)
	del rev
	assert bool((back.view(n, 12, 20) == parents.view(n, 1, 20)).all())
	assert int(fl.sum()) == stats[0].item() >= 2 and stats[1].item() == int(torch.nonzero(fl)[0])
	ch2, _ = cube.device.expand12(parents, want_flags=False)
	assert torch.equal(ch2, ch)


def test_device_api_rejects_bad_tensors():
	"""Raw pointers go to the C ABI: strided, wrongly typed or host tensors must be refused, not reinterpreted."""
	p = dev(random_walk(64, 5, seed=1))
	with pytest.raises(ValueError):
		cube.device.expand12(p[::2])                                   # strided view
	with pytest.raises(ValueError):
		cube.device.expand12(p.to(torch.int32))
	with pytest.raises(ValueError):
		cube.device.expand12(p.cpu())
	with pytest.raises(ValueError):
		cube.device.multi_rotate(p, torch.zeros(64, dtype=torch.int64, device="cuda"))
	with pytest.raises(ValueError):
		cube.device.multi_rotate(p, torch.zeros(63, dtype=torch.uint8, device="cuda"))
	with pytest.raises(ValueError):
		cube.device.expand12(p, children=torch.empty((100, 20), dtype=torch.int8, device="cuda"))
	with pytest.raises(ValueError):
		cube.device.as_oh(p, out=torch.empty((64, 480), dtype=torch.float16, device="cuda"))
	# the drop-in surface accepts strided / wider-typed input by making a dense int8 copy
	assert (cube.multi_rotate(p[::2].cpu().numpy().astype(np.int64), np.zeros(32, int), np.ones(32, int))
	        == orc.multi_rotate(p[::2].cpu().numpy(), np.zeros(32, int), np.ones(32, int))).all()


# ------------------------------------------------------------------------------------------------- the paced fan-out
_PACED_CHILD = r"""
import sys, torch
from librubiks_amd import _ffi, cube
n = int(sys.argv[1])
assert _ffi.lib().rk_init(0) == 0
g = torch.Generator(device="cuda"); g.manual_seed(11)
parents = cube.device.apply_sequences(torch.randint(0, 12, (7, n), device="cuda", dtype=torch.uint8, generator=g), False, True)
solved = torch.from_numpy(cube.get_solved()).cuda()
for i, a in ((0, 0), (n // 2, 5), (n - 1, 10)):                       # three parents one move from the goal
	parents[i] = cube.device.multi_rotate(solved[None], torch.tensor([a], dtype=torch.uint8, device="cuda"))[0]
stats = torch.tensor([0, _ffi.INT64_MAX], dtype=torch.int64, device="cuda")
ch, fl = cube.device.expand12(parents, stats=stats)
# row by row with the per-state kernels: child 12 i + a = parent i turned by a
ref = cube.device.multi_rotate(parents.repeat_interleave(12, 0), torch.arange(12, dtype=torch.uint8, device="cuda").repeat(n))
assert torch.equal(ch, ref), "children"
ref_fl = cube.device.multi_is_solved(ref)
assert torch.equal(fl, ref_fl), "flags"
assert int(ref_fl.sum()) == stats[0].item() >= 3 and stats[1].item() == int(torch.nonzero(ref_fl)[0]), "stats"
ch2, none = cube.device.expand12(parents, want_flags=False)
assert none is None and torch.equal(ch2, ref), "children without flags"
print("paced ok", n)
"""


@pytest.mark.parametrize("env", [
	{"RK_PACE_MIN": "1", "RK_PACE_PHASE": "4096", "RK_PACE_PULL": "64"},      # several short phases
	{"RK_PACE_MIN": "1", "RK_PACE_PHASE": "4096", "RK_PACE_PULL": "0"},       # no read phase: the first wave of a phase sets the time base
	{"RK_PACE_MIN": "1", "RK_PACE_TAU_PS": "20000", "RK_PACE_LEAD": "1000"},  # a schedule far slower than the memory: every wave waits
	{"RK_PACE_MIN": "1", "RK_PACE_TAU_PS": "100"},                            # a schedule nobody can keep: every wave is behind it
	{"RK_PACE": "0"},                                                         # the unpaced forms
])
@pytest.mark.parametrize("n", [300_003, 4096 * 64 * 2])
def test_paced_fanout_in_every_shape(env, n):
	"""
	The paced form of the fan-out (a read phase, then one tile per wave stored on a schedule) decides only WHEN a
	finished tile is stored.  Its constants are read from the environment once per process, so every shape runs in a process
	of its own: phases that end inside the batch and exactly at its end, a ragged last tile, no read phase, a schedule far
	too slow and one far too fast -- children, flags and statistics equal the per-state kernels' every time.
	"""
	import os
	import subprocess
	import sys
	root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
	out = subprocess.run([sys.executable, "-c", _PACED_CHILD, str(n)], env={**os.environ, **env}, cwd=root, capture_output=True, text=True, timeout=300)
	assert out.returncode == 0 and f"paced ok {n}" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_paced_fanout_replayed_from_a_hipgraph_and_on_two_streams():
	"""Nothing of the paced launch lives on the host: a captured launch replays, and two launches in flight at once (they share
	the device-side time base, so each may shift the other's schedule by microseconds) still write the right children."""
	n = 400_000
	g = torch.Generator(device="cuda")
	g.manual_seed(13)
	a = cube.device.apply_sequences(torch.randint(0, 12, (9, n), device="cuda", dtype=torch.uint8, generator=g), False, True)
	b = cube.device.apply_sequences(torch.randint(0, 12, (9, n), device="cuda", dtype=torch.uint8, generator=g), False, True)
	ref_a, fl_a = cube.device.expand12(a)
	ref_b, fl_b = cube.device.expand12(b)
	src = a.clone()
	ch, fl = torch.empty_like(ref_a), torch.empty_like(fl_a)
	graph = torch.cuda.CUDAGraph()
	s = torch.cuda.Stream()
	with torch.cuda.stream(s):
		cube.device.expand12(src, ch, fl)                               # warm-up outside the capture
		torch.cuda.synchronize()
		with torch.cuda.graph(graph, stream=s):
			cube.device.expand12(src, ch, fl)
	for want_ch, want_fl, inp in ((ref_a, fl_a, a), (ref_b, fl_b, b), (ref_a, fl_a, a)):
		src.copy_(inp); ch.zero_(); fl.fill_(7)
		graph.replay()
		torch.cuda.synchronize()
		assert torch.equal(ch, want_ch) and torch.equal(fl, want_fl)
	s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
	out1, out2 = (torch.empty_like(ref_a), torch.empty_like(fl_a)), (torch.empty_like(ref_b), torch.empty_like(fl_b))
	torch.cuda.synchronize()
	for _ in range(3):
		with torch.cuda.stream(s1):
			cube.device.expand12(a, *out1)
		with torch.cuda.stream(s2):
			cube.device.expand12(b, *out2)
	torch.cuda.synchronize()
	assert torch.equal(out1[0], ref_a) and torch.equal(out1[1], fl_a) and torch.equal(out2[0], ref_b) and torch.equal(out2[1], fl_b)


@pytest.mark.parametrize("n", [40_003, 70_001])
def test_repr686_paced_fanout(n):
	"""From 65 536 parents on the 6x8x6 fan-out runs in the paced form (k_fanout686p: eight parents per workgroup, read phases,
	stores on a schedule; 70 001 parents are two phases and a ragged last group), below that on the persistent grid (40 003):
	children against the per-state kernel on device, a sample and the planted solved children against the oracle, flags and
	statistics against the goal-test kernel."""
	cube.set_is2024(False)
	try:
		g = torch.Generator(device="cuda")
		g.manual_seed(686 + n)
		p = dev(np.broadcast_to(orc.SOLVED686, (n, 6, 8, 6)).copy())
		for _ in range(6):
			p = cube.device.multi_rotate(p, torch.randint(0, 12, (n,), device="cuda", dtype=torch.uint8, generator=g))
		for i, (face, d) in ((0, (4, 0)), (n // 2, (2, 1)), (n - 1, (0, 1))):
			p[i] = dev(orc.rotate686(orc.SOLVED686, face, d))
		stats = torch.tensor([0, _ffi.INT64_MAX], dtype=torch.int64, device="cuda")
		ch, fl = cube.device.expand12(p, stats=stats)
		ref = cube.device.multi_rotate(p.repeat_interleave(12, 0), torch.arange(12, dtype=torch.uint8, device="cuda").repeat(n))
		assert torch.equal(ch, ref)
		ref_fl = cube.device.multi_is_solved(ref)
		assert torch.equal(fl, ref_fl) and int(ref_fl.sum()) == stats[0].item() >= 3 and stats[1].item() == int(torch.nonzero(ref_fl)[0])
		pick = [0, 1, n // 2, n - 2, n - 1, 16384 * 4 - 1 if n > 16384 * 4 else 5, min(n - 1, 16384 * 4)]
		pn = p[pick].cpu().numpy()
		want = np.stack([orc.rotate686(s, a // 2, 1 - a % 2) for s in pn for a in range(12)])
		assert (ch.view(n, 12, 6, 8, 6)[pick].reshape(-1, 6, 8, 6).cpu().numpy() == want).all()
		ch2, none = cube.device.expand12(p, want_flags=False)
		assert none is None and torch.equal(ch2, ref)
	finally:
		cube.set_is2024(True)


def test_paced_multi_rotate_large_batch():
	"""From 2 Mi states on multi_rotate runs with its loads released on a schedule and non-temporal stores (one tile per wave):
	same results as below that size -- a sample against the oracle, everything against the inverse move, in place, with a
	ragged last tile, and through the faces / directions entry."""
	n = 2_100_003
	g = torch.Generator(device="cuda")
	g.manual_seed(21)
	states = cube.device.apply_sequences(torch.randint(0, 12, (15, n), device="cuda", dtype=torch.uint8, generator=g), False, True)
	acts = torch.randint(0, 12, (n,), device="cuda", dtype=torch.uint8, generator=g)
	out = cube.device.multi_rotate(states, acts)
	pick = torch.cat([torch.arange(0, 2048), torch.randint(0, n, (4096,)), torch.arange(n - 2048, n)]).cuda()
	ref = c_oracle.multi_rotate(states[pick].cpu().numpy(), acts[pick].cpu().numpy())
	assert (out[pick].cpu().numpy() == ref).all()
	assert torch.equal(cube.device.multi_rotate(out, acts ^ 1), states)
	small = torch.cat([cube.device.multi_rotate(states[i:i + 700_001], acts[i:i + 700_001]) for i in range(0, n, 700_001)])   # unpaced launches
	assert torch.equal(small, out)
	inplace = states.clone()
	assert cube.device.multi_rotate(inplace, acts, inplace) is inplace and torch.equal(inplace, out)
	faces, dirs = (acts // 2).cpu().numpy(), (1 - acts % 2).cpu().numpy()
	assert torch.equal(cube.multi_rotate(states, faces, dirs), out)


def test_paced_fanout_misaligned_views():
	"""The paced fan-out on views that are only 4-byte aligned (parents one row into a tensor: the read phase needs 16-byte
	alignment and is skipped) and with a flag buffer four bytes into a tensor (the 768-byte flag blocks go out as dwords)."""
	n = 250_000
	g = torch.Generator(device="cuda")
	g.manual_seed(17)
	base = cube.device.apply_sequences(torch.randint(0, 12, (6, n + 1), device="cuda", dtype=torch.uint8, generator=g), False, True)
	parents = base[1:]
	assert parents.data_ptr() % 16 != 0
	ref_ch, ref_fl = cube.device.expand12(parents.clone())
	flag_store = torch.empty(12 * n + 4, dtype=torch.uint8, device="cuda")
	ch, fl = cube.device.expand12(parents, solved=flag_store[4:])
	assert torch.equal(ch, ref_ch) and torch.equal(fl, ref_fl)
	with pytest.raises(_ffi.RubiksHipError, match="4-byte aligned"):                     # refused loudly, not mis-stored
		cube.device.expand12(parents, solved=flag_store[1:12 * n + 1])
	pick = torch.randint(0, n, (2048,)).cuda()
	want_ch, want_fl = c_oracle.expand12(parents[pick].cpu().numpy())
	assert (ref_ch.view(n, 12, 20)[pick].reshape(-1, 20).cpu().numpy() == want_ch).all()
	assert (ref_fl.view(n, 12)[pick].reshape(-1).cpu().numpy() == want_fl).all()


def test_bad_action_codes_leave_a_mark():
	"""Device-pointer entries do not validate action codes (no reduction, no synchronisation per call): a code >= 12 acts as
	action 0 and sets a sticky mark that `cube.device.bad_actions_seen()` reads and clears -- on the aligned fast path, on the
	ragged path, in the scrambler and in the 6x8x6 move kernel."""
	g = torch.Generator(device="cuda")
	g.manual_seed(3)
	cube.device.bad_actions_seen()
	for n, pos, code in ((1024, 700, 200), (1024, 3, 12), (1024, 1023, 15), (1001, 1000, 12), (777, 0, 255)):
		states = cube.device.apply_sequences(torch.randint(0, 12, (5, n), device="cuda", dtype=torch.uint8, generator=g), False, True)
		acts = torch.randint(0, 12, (n,), device="cuda", dtype=torch.uint8, generator=g)
		acts[:12] = torch.arange(12, dtype=torch.uint8, device="cuda")      # every valid code, 8..11 included
		good = cube.device.multi_rotate(states, acts)
		assert not cube.device.bad_actions_seen()
		bad = acts.clone()
		bad[pos] = code
		out = cube.device.multi_rotate(states, bad)
		zero = acts.clone()
		zero[pos] = 0
		assert torch.equal(out, cube.device.multi_rotate(states, zero))        # treated as action 0
		assert cube.device.bad_actions_seen() and not cube.device.bad_actions_seen()
		assert torch.equal(cube.device.multi_rotate(states, acts), good) and not cube.device.bad_actions_seen()
	seq = torch.randint(0, 12, (6, 300), device="cuda", dtype=torch.uint8, generator=g)
	seq[3, 17] = 12
	cube.device.apply_sequences(seq, False, True)
	assert cube.device.bad_actions_seen()
	cube.set_is2024(False)
	try:
		s6 = dev(np.broadcast_to(orc.SOLVED686, (500, 6, 8, 6)).copy())
		a6 = torch.randint(0, 12, (500,), device="cuda", dtype=torch.uint8, generator=g)
		cube.device.multi_rotate(s6, a6)
		assert not cube.device.bad_actions_seen()
		a6[499] = 13
		cube.device.multi_rotate(s6, a6)
		assert cube.device.bad_actions_seen()
	finally:
		cube.set_is2024(True)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_paced_as_oh_every_dtype(dtype):
	"""From 8 192 tiles on (65 536 states for f32, 131 072 for the 16-bit types) `as_oh` runs in the paced form (read phases,
	non-temporal stores on a schedule).  300 007 states: a ragged last tile; two phases for bf16/f16 need more than 1 Mi
	states, which the second size provides.  Against the index of the one and against the unpaced kernel on slices."""
	g = torch.Generator(device="cuda")
	g.manual_seed(480)
	for n in (300_007, 1_100_003):
		states = cube.device.apply_sequences(torch.randint(0, 12, (9, n), device="cuda", dtype=torch.uint8, generator=g), False, True)
		oh = cube.device.as_oh(states, dtype=dtype)
		assert oh.shape == (n, 480) and oh.dtype == dtype
		idx = (states.to(torch.int64) + 24 * torch.arange(20, device="cuda")).reshape(n, 20)
		assert bool((oh.view(n, 20, 24).argmax(-1).reshape(n, 20) + 24 * torch.arange(20, device="cuda") == idx).all())
		assert float(oh.float().sum()) == 20.0 * n
		for lo in (0, n // 2, n - 5000):
			assert torch.equal(cube.device.as_oh(states[lo:lo + 5000], dtype=dtype), oh[lo:lo + 5000])      # unpaced launches
		del oh


@pytest.mark.parametrize("n", [1, 3, 63, 256, 257, 1000, 70_001, 2_100_003])
def test_multi_rotate_solved_is_the_two_calls_in_one(n):
	"""rk_multi_rotate_solved: `multi_rotate` and `multi_is_solved` of the moved states in one launch -- the pair of
	agents.py:157-159, :696-703, train.py:277-281.  Moved states, flags, count and first index against the two separate kernels and
	(a sample) the C oracle; planted rows one move from solved; in place; ragged tiles; the paced size (2 Mi states on)."""
	g = torch.Generator(device="cuda")
	g.manual_seed(1000 + n % 997)
	states = cube.device.apply_sequences(torch.randint(0, 12, (11, n), device="cuda", dtype=torch.uint8, generator=g), False, True)
	acts = torch.randint(0, 12, (n,), device="cuda", dtype=torch.uint8, generator=g)
	plant = sorted({0, n // 3, n - 1})
	for k, i in enumerate(plant):                                              # state = rev(a) of solved, action = a: the move solves it
		a = (5 * k + 2) % 12
		states[i] = dev(orc.rotate(orc.SOLVED, (a ^ 1) // 2, 1 - (a ^ 1) % 2))
		acts[i] = a
	want = cube.device.multi_rotate(states, acts)
	want_fl = cube.device.multi_is_solved(want)
	stats = torch.tensor([0, _ffi.INT64_MAX], dtype=torch.int64, device="cuda")
	out, fl = cube.device.multi_rotate_solved(states, acts, stats=stats)
	assert torch.equal(out, want) and torch.equal(fl, want_fl)
	assert int(fl.sum()) == stats[0].item() >= len(plant) and stats[1].item() == int(torch.nonzero(fl)[0]) == 0
	assert all(bool(fl[i]) for i in plant)
	pick = torch.unique(torch.cat([torch.arange(0, min(n, 512)), torch.randint(0, n, (min(n, 2048),)), torch.arange(max(0, n - 512), n)])).cuda()
	ref = c_oracle.multi_rotate(states[pick].cpu().numpy(), acts[pick].cpu().numpy())
	assert (out[pick].cpu().numpy() == ref).all() and (fl[pick].cpu().numpy().astype(bool) == orc.multi_is_solved(ref)).all()
	# in place, flags only / stats only, misaligned flag buffer (byte path)
	inplace = states.clone()
	store = torch.zeros(n + 3, dtype=torch.uint8, device="cuda")
	o2, f2 = cube.device.multi_rotate_solved(inplace, acts, out=inplace, flags=store[3:])
	assert o2 is inplace and torch.equal(inplace, want) and torch.equal(f2, want_fl)
	lib = _ffi.lib()
	st2 = torch.tensor([0, _ffi.INT64_MAX], dtype=torch.int64, device="cuda")
	tmp = torch.empty_like(states)
	_ffi.check(lib.rk_multi_rotate_solved(_ffi.REPR_2024, states.data_ptr(), acts.data_ptr(), tmp.data_ptr(), None, st2.data_ptr(), n, None))
	assert torch.equal(tmp, want) and st2.tolist() == stats.tolist()
	assert lib.rk_multi_rotate_solved(_ffi.REPR_2024, states.data_ptr(), acts.data_ptr(), tmp.data_ptr(), None, None, n, None) == -1


def test_multi_rotate_solved_686():
	cube.set_is2024(False)
	try:
		n = 3001
		g = torch.Generator(device="cuda")
		g.manual_seed(9)
		p = dev(np.broadcast_to(orc.SOLVED686, (n, 6, 8, 6)).copy())
		for _ in range(4):
			p = cube.device.multi_rotate(p, torch.randint(0, 12, (n,), device="cuda", dtype=torch.uint8, generator=g))
		acts = torch.randint(0, 12, (n,), device="cuda", dtype=torch.uint8, generator=g)
		p[17] = dev(orc.rotate686(orc.SOLVED686, 2, 1))
		acts[17] = 2 * 2 + 1                                                # (face 2, dir 0) undoes (face 2, dir 1)
		want = cube.device.multi_rotate(p, acts)
		stats = torch.tensor([0, _ffi.INT64_MAX], dtype=torch.int64, device="cuda")
		out, fl = cube.device.multi_rotate_solved(p, acts, stats=stats)
		assert torch.equal(out, want) and torch.equal(fl, cube.device.multi_is_solved(want)) and bool(fl[17]) and stats[0].item() == int(fl.sum())
	finally:
		cube.set_is2024(True)


def test_paced_kernels_of_different_kinds_on_two_streams():
	"""VERDICT r3 #7 / ADVICE r3: every paced launch has its own time base now (a cell of g_pace_cells handed out per launch), so
	two paced kernels of DIFFERENT kinds in flight on two streams -- what the sharded search and ADI run -- cannot move each
	other's schedule.  Results never depended on the base; they are checked here for the pairs fan-out / as_oh and fan-out /
	6x8x6 fan-out, several rounds each (benchmarks/pace_streams.py records what the concurrency costs in time)."""
	n = 600_000
	g = torch.Generator(device="cuda")
	g.manual_seed(77)
	a = cube.device.apply_sequences(torch.randint(0, 12, (9, n), device="cuda", dtype=torch.uint8, generator=g), False, True)
	b = cube.device.apply_sequences(torch.randint(0, 12, (9, n), device="cuda", dtype=torch.uint8, generator=g), False, True)
	ref_ch, ref_fl = cube.device.expand12(a)
	ref_oh = cube.device.as_oh(b, dtype=torch.bfloat16)
	s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
	ch, fl, oh = torch.empty_like(ref_ch), torch.empty_like(ref_fl), torch.empty_like(ref_oh)
	torch.cuda.synchronize()
	for _ in range(4):
		ch.zero_(); fl.fill_(9); oh.zero_()
		torch.cuda.synchronize()
		with torch.cuda.stream(s1):
			cube.device.expand12(a, ch, fl)
			cube.device.expand12(a, ch, fl)
		with torch.cuda.stream(s2):
			cube.device.as_oh(b, oh, torch.bfloat16)
			cube.device.as_oh(b, oh, torch.bfloat16)
		torch.cuda.synchronize()
		assert torch.equal(ch, ref_ch) and torch.equal(fl, ref_fl) and torch.equal(oh, ref_oh)
	del oh, ref_oh
	cube.set_is2024(False)
	try:
		m = 80_000
		p = dev(np.broadcast_to(orc.SOLVED686, (m, 6, 8, 6)).copy())
		for _ in range(5):
			p = cube.device.multi_rotate(p, torch.randint(0, 12, (m,), device="cuda", dtype=torch.uint8, generator=g))
		ref6, ref6_fl = cube.device.expand12(p)
		ch6, fl6 = torch.empty_like(ref6), torch.empty_like(ref6_fl)
		for _ in range(3):
			ch.zero_(); ch6.zero_()
			torch.cuda.synchronize()
			with torch.cuda.stream(s1):
				cube.set_is2024(True)
				cube.device.expand12(a, ch, fl)
			with torch.cuda.stream(s2):
				cube.set_is2024(False)
				cube.device.expand12(p, ch6, fl6)
			torch.cuda.synchronize()
			assert torch.equal(ch, ref_ch) and torch.equal(ch6, ref6) and torch.equal(fl6, ref6_fl)
	finally:
		cube.set_is2024(True)


def test_set_pacing_switches_the_form_not_the_results():
	"""rk_set_pacing(0 / 1 / -1): the unpaced kernels at every size, the paced forms, back to the environment's choice -- at run time,
	which is how bench.py times both forms in one process.  Same children, flags and one-hot either way."""
	lib = _ffi.lib()
	n = 400_000
	g = torch.Generator(device="cuda")
	g.manual_seed(33)
	p = cube.device.apply_sequences(torch.randint(0, 12, (8, n), device="cuda", dtype=torch.uint8, generator=g), False, True)
	try:
		outs = []
		for mode in (1, 0, -1, 0, 1):
			_ffi.check(lib.rk_set_pacing(mode))
			ch, fl = cube.device.expand12(p)
			oh = cube.device.as_oh(p, dtype=torch.bfloat16)
			outs.append((ch, fl, oh))
		for ch, fl, oh in outs[1:]:
			assert torch.equal(ch, outs[0][0]) and torch.equal(fl, outs[0][1]) and torch.equal(oh, outs[0][2])
	finally:
		_ffi.check(lib.rk_set_pacing(-1))


# ------------------------------------------------------------------------------------------------- librubiks.cube.maps names
def test_as_oh_layout_through_the_maps_helpers():
	"""The body of ref:tests/test_cube.py:129-139: the one-hot of the solved state, rebuilt from SimpleState through
	get_corner_pos / get_side_pos, with the helpers imported the way the reference's test imports them."""
	from librubiks_amd import gpu
	from librubiks_amd.cube.maps import SimpleState, get_corner_pos, get_side_pos
	state = cube.get_solved()
	oh = cube.as_oh(state)
	supposed_state = torch.zeros(20, 24, device=gpu)
	corners = [get_corner_pos(c, o) for c, o in zip(SimpleState.corners.tolist(), SimpleState.corner_orientations.tolist())]
	supposed_state[torch.arange(8), corners] = 1
	sides = [get_side_pos(s, o) for s, o in zip(SimpleState.sides.tolist(), SimpleState.side_orientations.tolist())]
	supposed_state[torch.arange(8, 20), sides] = 1
	assert (supposed_state.flatten() == oh).all()


def test_tensor_map_moves_states_like_the_kernels():
	"""`state + maps[dir, face][kind, state]` (ref:cube/cube.py:244-254) with the drop-in's get_tensor_map = the device's rotate."""
	maps = cube.get_tensor_map(np.int8)
	kind = np.array([0] * 8 + [1] * 12)
	s = random_walk(64, 15, seed=5)
	for face in range(6):
		for d in range(2):
			want = s + maps[d, face][kind, s]
			assert np.array_equal(cube.multi_rotate(s, np.full(64, face), np.full(64, d)), want)


def test_host_entries_refuse_a_single_state_as_rows():
	"""len() of a (20,) state is 20: the zero-copy entries must not read 20 rows from a 20-byte buffer (advisor, round 4)."""
	one = cube.get_solved()
	with pytest.raises(ValueError):
		cube.expand(one)
	with pytest.raises(ValueError):
		cube.multi_is_solved(one)
	with pytest.raises(ValueError):
		cube.multi_rotate(one, np.zeros(20, np.uint8), np.zeros(20, np.uint8))
	assert cube.is_solved(one) and cube.expand(one[None]).shape == (12, 20)


# ------------------------------------------------------------------------------------------------- pacing: calibration, stream lifetime
def _get_pacing():
	tau, src, us = C.c_uint(0), C.c_int(-1), (C.c_float * 5)()
	_ffi.check(_ffi.lib().rk_get_pacing(C.byref(tau), C.byref(src), us))
	return tau.value, src.value, list(us)


def test_calibrated_schedule_is_one_of_the_candidates_and_changes_no_result():
	"""rk_init measures the store schedule once per device (rk_calibrate_pacing / rk_get_pacing): the schedule in force is one of
	the candidates (0 = the ring form), every candidate was timed, the compiled 2.1 ns stay unless something beats them by 3 % --
	and whatever was chosen, children and flags are the oracle's."""
	import os
	lib = _ffi.lib()
	tau, src, us = _get_pacing()
	if os.environ.get("RK_PACE_TAU_PS") or os.environ.get("RK_PACE", "1") == "0" or os.environ.get("RK_PACE_CALIBRATE", "1") == "0":
		assert src == 2
		return
	assert src == 1 and tau in (0, 2000, 2100, 2200, 2400)
	assert all(5.0 < u < 500.0 for u in us), us                       # 1 Mi parents: tens of microseconds per launch, every candidate timed
	if tau != 2100:
		assert us[[0, 2000, 2100, 2200, 2400].index(tau)] * 1.03 < us[2]
	_ffi.check(lib.rk_calibrate_pacing(1))                             # again, forced: still a candidate, still timed
	tau2, src2, us2 = _get_pacing()
	assert src2 == 1 and tau2 in (0, 2000, 2100, 2200, 2400) and all(u > 0 for u in us2)
	n = 300_000
	p = random_walk_c(n, 12, seed=77)
	ch, fl = cube.device.expand12(dev(p))
	want, want_fl = c_oracle.expand12(p, threads=8)
	assert np.array_equal(ch.cpu().numpy(), want) and np.array_equal(fl.cpu().numpy(), want_fl)


def test_a_stream_destroyed_between_two_paced_launches():
	"""VERDICT r4 #6 / advisor: the turn-taking of paced launches remembers the previous paced launch's stream -- but only a stream
	it may rely on: the null stream or a REGISTERED one (rk_stream_register: alive until rk_stream_forget).  (1) A caller's own stream,
	never registered, runs a paced launch and is destroyed: nothing of it was kept, the next paced launch on torch's stream neither
	faults nor waits.  (2) Registered, used, forgotten, destroyed: the same.  (3) Two registered live streams do take turns and both
	results are right.  On this HIP runtime touching a destroyed stream's handle is a segmentation fault, so a fault here IS the test."""
	hip = C.CDLL("libamdhip64.so")
	hip.hipStreamCreate.argtypes, hip.hipStreamDestroy.argtypes, hip.hipStreamSynchronize.argtypes = [C.POINTER(C.c_void_p)], [C.c_void_p], [C.c_void_p]
	lib = _ffi.lib()
	n = 262_144                                                         # above the paced form's threshold (196 608 parents)
	p = random_walk_c(n, 10, seed=78)
	want, _ = c_oracle.expand12(p, threads=8)
	d_p = dev(p)
	children = torch.empty((12 * n, 20), dtype=torch.int8, device="cuda")
	flags = torch.empty(12 * n, dtype=torch.uint8, device="cuda")
	torch.cuda.synchronize()
	for register in (False, True):
		s = C.c_void_p()
		assert hip.hipStreamCreate(C.byref(s)) == 0
		if register:
			_ffi.check(lib.rk_stream_register(s))
		_ffi.check(lib.rk_expand12(_ffi.REPR_2024, d_p.data_ptr(), children.data_ptr(), flags.data_ptr(), None, n, s))       # paced, on the caller's own stream
		assert hip.hipStreamSynchronize(s) == 0
		assert np.array_equal(children.cpu().numpy(), want)
		if register:
			_ffi.check(lib.rk_stream_forget(s))
		assert hip.hipStreamDestroy(s) == 0
		ch, fl = cube.device.expand12(d_p)                              # paced, on torch's stream
		torch.cuda.synchronize()
		assert np.array_equal(ch.cpu().numpy(), want)
	# two live registered streams: the second launch waits for the first (an event on the first's stream), results are both right
	a, b = C.c_void_p(), C.c_void_p()
	assert hip.hipStreamCreate(C.byref(a)) == 0 and hip.hipStreamCreate(C.byref(b)) == 0
	_ffi.check(lib.rk_stream_register(a)); _ffi.check(lib.rk_stream_register(b))
	children2 = torch.zeros_like(children)
	torch.cuda.synchronize()
	_ffi.check(lib.rk_expand12(_ffi.REPR_2024, d_p.data_ptr(), children.data_ptr(), flags.data_ptr(), None, n, a))
	_ffi.check(lib.rk_expand12(_ffi.REPR_2024, d_p.data_ptr(), children2.data_ptr(), flags.data_ptr(), None, n, b))
	assert hip.hipStreamSynchronize(b) == 0 and hip.hipStreamSynchronize(a) == 0
	assert torch.equal(children, children2) and np.array_equal(children2.cpu().numpy(), want)
	for s in (a, b):
		_ffi.check(lib.rk_stream_forget(s))
		assert hip.hipStreamDestroy(s) == 0
	ch, _ = cube.device.expand12(d_p)
	torch.cuda.synchronize()
	assert np.array_equal(ch.cpu().numpy(), want)
	assert lib.rk_stream_forget(None) == 0 and lib.rk_stream_register(None) == 0


# ------------------------------------------------------------------------------------------------- ADI rollout in one launch
@pytest.mark.parametrize("games,depth,with_solved", [(1, 1, False), (1, 1, True), (3, 7, True), (5, 13, False), (64, 1, False), (77, 30, True), (7500, 30, False),
                                                      (9, 64, True), (11, 33, False), (70, 2, True), (3, 70, True), (2, 100, False), (1000, 3, False)])
def test_rollout_fanout_is_walk_plus_goal_test_plus_fanout(games, depth, with_solved):
	"""rk_rollout_fanout (VERDICT r4 #4b): the states along every game's walk, their goal test, their 12 children and the children's
	goal test in one launch -- against the oracle's sequence_scrambler / fan-out on the same draws (ref:cube/cube.py:218-232,
	ref:train.py:277-292) and against the three launches it replaces.  Sizes around the 64-state tiles, up to the reference's
	rollout (7 500 games x 30); games of up to 64 rows go through the prefix scan over the moves' permutations (whole games per wave:
	1, 2, 3, 7, 13, 30, 33, 64 rows), longer ones through the per-lane walk (70, 100 rows)."""
	rng = np.random.RandomState(1000 * games + depth)
	faces, dirs = rng.randint(0, 6, (depth, games)), rng.randint(0, 2, (depth, games))
	acts = dev((2 * faces + (1 - dirs)).astype(np.uint8))
	states, sfl, children, cfl = cube.device.rollout_fanout(acts, with_solved)
	# the oracle's walk: game-major rows, the solved state in front when asked (and then only depth - 1 moves)
	cur = orc.repeat_state(orc.SOLVED, games)
	seq = [cur] if with_solved else []
	for d in range(depth - int(with_solved)):
		cur = orc.multi_rotate(cur, faces[d], dirs[d])
		seq.append(cur)
	want = np.stack(seq, axis=1).reshape(games * depth, 20)
	assert np.array_equal(states.cpu().numpy(), want)
	assert np.array_equal(sfl.cpu().numpy().astype(bool), orc.multi_is_solved(want))
	want_ch, want_fl = c_oracle.expand12(want, threads=8)
	assert np.array_equal(children.cpu().numpy(), want_ch) and np.array_equal(cfl.cpu().numpy(), want_fl)
	if with_solved:
		assert sfl[::depth].all() and int(cfl.sum()) >= 0
	# ... and the three launches it replaces
	s3 = cube.device.apply_sequences(acts, with_solved, False)
	ch3, fl3 = cube.device.expand12(s3)
	assert torch.equal(s3, states) and torch.equal(ch3, children) and torch.equal(fl3, cfl) and torch.equal(cube.device.multi_is_solved(s3), sfl)


@pytest.mark.parametrize("games,depth,with_solved,only_last", [(1, 999, False, True), (1, 100, False, False), (3, 64, True, False), (5, 65, True, False),
                                                               (7, 129, False, False), (300, 2, True, False), (1024, 9, False, True), (1025, 9, False, True), (1024, 8, True, False),
                                                               (2, 64, False, True), (1, 2, True, True), (16, 7, False, False), (4, 200, True, True)])
def test_scramblers_as_scans_over_moves(games, depth, with_solved, only_last):
	"""Few games: a wave per game, the moves of a 64-move chunk composed by a prefix scan over their permutation tables, the last state
	carried into the next chunk (k_apply_sequences_scan) -- deep single scrambles (the evaluation loop's depth 100-999), chunk boundaries
	(64, 65, 129 moves), the solved row in front, both sides of the switch to the lane-per-game walk (1 024 / 1 025 games, 7 / 8 moves): every
	row equals the oracle's walk on the same draws."""
	rng = np.random.RandomState(31 * games + depth)
	faces, dirs = rng.randint(0, 6, (depth, games)), rng.randint(0, 2, (depth, games))
	acts = dev((2 * faces + (1 - dirs)).astype(np.uint8))
	got = cube.device.apply_sequences(acts, with_solved, only_last).cpu().numpy()
	cur = orc.repeat_state(orc.SOLVED, games)
	seq = [cur] if with_solved else []
	for d in range(depth - int(with_solved)):
		cur = c_oracle.multi_rotate(cur, (2 * faces[d] + (1 - dirs[d])).astype(np.uint8), threads=4)
		seq.append(cur)
	want = seq[-1] if only_last else np.stack(seq, axis=1).reshape(games * depth, 20)
	assert np.array_equal(got, want)
	if games == 1 and only_last:                                             # ... and through the drop-in: scramble(depth) on the reference's draws
		np.random.seed(depth)
		s, f, dr = cube.scramble(depth)
		np.random.seed(depth)
		s_ref, _, _ = orc.scramble(depth)
		assert np.array_equal(s, s_ref)
