"""
Pins the CPU oracle (oracle/cube_oracle.py and oracle/cube_oracle.c) to the golden vectors captured from the
real reference (oracle/gen_golden.py) and to the data the reference ships itself (frontend maps.json, the
literals of its tests/test_cube.py).  CPU only.
"""
import numpy as np
import pytest

from oracle import c_oracle, cube_oracle as orc
from tests.helpers import random_walk, random_walk_c, sha


def test_tables_match_reference_and_frontend(golden):
	t = golden["cube_tables"]
	assert orc.DELTA.dtype == np.int8 and (orc.DELTA == t["delta_maps"]).all()
	# independent vector: the Angular frontend's maps.json (frontend/src/assets/maps.json)
	assert (orc.DELTA[0] == t["frontend_map_neg"]).all()
	assert (orc.DELTA[1] == t["frontend_map_pos"]).all()
	assert (orc.SOLVED == t["solved2024"]).all()
	assert (orc.SOLVED == np.array([0, 3, 6, 9, 12, 15, 18, 21, 0, 2, 4, 6, 8, 10, 12, 14, 16, 18, 20, 22])).all()  # cube.service.ts:33
	assert (orc.SOLVED686 == t["solved686"]).all()
	assert [tuple(int(v) for v in x) for x in t["action_space"]] == [tuple(int(v) for v in orc.action_face_dir(a)) for a in range(12)]
	# hashes recorded in SURVEY.md 8c
	assert sha(orc.DELTA) == "a24fd07addac723f3f9e1f84e1136bf76cb28348a2677035482712f52ea02827"
	assert sha(orc.LUT.astype(np.int8)) == "4b1e5a2445714895c679f5712d1f3bdc4526efe02e4949a8c5930d8e94f21ed4"
	assert sha(orc.PERM686) == "86c2e05c09f013f4de60fe1e8a2a951684e9e5827d91b3bb36c9e6e03b326a5a"


def test_lut_structure():
	for a in range(12):
		for k, moved in ((0, 12), (1, 8)):
			p = orc.LUT[a, k]
			assert sorted(p) == list(range(24))
			assert (p != np.arange(24)).sum() == moved
			assert (p[p[p[p]]] == np.arange(24)).all()          # order 4
			assert (orc.LUT[a ^ 1, k][p] == np.arange(24)).all()  # rev action is the inverse
	lut_c, perm_c, solved_c = c_oracle.tables()
	assert (lut_c == orc.LUT).all() and (perm_c == orc.PERM686).all() and (solved_c == orc.SOLVED).all()


def test_scramble_known_answers(golden):
	k = golden["cube_kat"]
	np.random.seed(0)
	s, f, d = orc.scramble(5)
	assert (s == k["k1_state"]).all() and (f == k["k1_faces"]).all() and (d == k["k1_dirs"]).all()
	assert s.tolist() == [13, 20, 4, 11, 7, 17, 22, 2, 2, 8, 6, 10, 18, 20, 22, 0, 16, 12, 4, 14]
	np.random.seed(42)
	s, f, d = orc.scramble(1)
	assert (s == k["k2a_state"]).all() and (f == k["k2a_faces"]).all() and (d == k["k2a_dirs"]).all()
	s, f, d = orc.scramble(20)
	assert (s == k["k2b_state"]).all()
	# tests/test_cube.py:103-114: undoing the scramble solves the cube
	for face, dr in zip(reversed(f), reversed(d)):
		s = orc.rotate(s, face, 1 - dr)
	assert orc.is_solved(s)
	np.random.seed(7)
	s, f, d = orc.scramble(6, True)
	assert (s == k["k7_state"]).all() and (f == k["k7_faces"]).all()


def test_fanout_and_rotate_known_answers(golden):
	k, text = golden["cube_kat"], golden["text"]
	p = random_walk(256, 20, seed=1)
	assert (p == k["k3_256_parents"]).all()
	assert (orc.expand12(p) == k["k3_256_children"]).all()
	ch_c, fl_c = c_oracle.expand12(p, threads=2)
	assert (ch_c == k["k3_256_children"]).all() and not fl_c.any()
	assert (orc.multi_rotate(p, k["mr_faces"], k["mr_dirs"]) == k["mr_out"]).all()
	acts = (2 * k["mr_faces"] + (1 - k["mr_dirs"])).astype(np.uint8)
	assert (c_oracle.multi_rotate(p, acts) == k["mr_out"]).all()
	# per-state rotate equals the batched form (tests/test_cube.py:94-101, but with both directions)
	for i in range(0, 256, 17):
		assert (orc.rotate(p[i], k["mr_faces"][i], k["mr_dirs"][i]) == k["mr_out"][i]).all()

	p = random_walk(10_000, 20, seed=1)
	assert sha(p) == text["k3_10k_parents_sha256"] == "8c27d09f852128276fef5b53a88f3f0c8c89f72ab330f19c94ca28e9f9c06502"
	c = orc.expand12(p)
	assert sha(c) == text["k3_10k_children_sha256"] == "be54dd06f609bc3b3a724e7739e8b5408d0cd78995df7f698ae16c8eb614d8ed"
	assert (p[0] == k["k3_10k_parent0"]).all() and (c[0] == k["k3_10k_child0"]).all()
	assert int(orc.multi_is_solved(c).sum()) == text["k3_10k_solved_children"] == 0
	# config 1 of BASELINE.json: all 12 faces, inverse move restores, solved round trip
	for a in range(12):
		f, d = np.full(len(p), a // 2), np.full(len(p), 1 - a % 2)
		assert (orc.multi_rotate(orc.multi_rotate(p, f, d), f, 1 - d) == p).all()
	s = orc.repeat_state(orc.SOLVED, 12)
	one = orc.multi_rotate(s, *orc.iter_actions())
	assert not orc.multi_is_solved(one).any()
	assert orc.multi_is_solved(orc.multi_rotate(one, orc.iter_actions()[0], 1 - orc.iter_actions()[1])).all()


def test_fanout_1m_hash_c_oracle(golden):
	"""The headline parity set (n = 1 M) through the C oracle: same SHA-256 as the reference produced."""
	import hashlib
	text = golden["text"]
	p = random_walk_c(1_000_000, 20, seed=1)
	assert sha(p) == text["k3_1m_parents_sha256"] == "f7424ff6657eff63080793adbdfc2da88088eced4c3da611e7db984aa1836791"
	h, nsolved = hashlib.sha256(), 0
	for lo in range(0, len(p), 250_000):
		c, fl = c_oracle.expand12(p[lo:lo + 250_000], threads=8)
		h.update(c.tobytes())
		nsolved += int(fl.sum())
	assert h.hexdigest() == text["k3_1m_children_sha256"] == "ef27a8d51abc0117ee198bf431ccb5456cd5dffd8122c329b1d246754064d64f"
	assert nsolved == text["k3_1m_solved_children"] == 0


def test_goal_test(golden):
	k = golden["cube_kat"]
	assert (orc.multi_is_solved(k["solved_mix"]) == k["solved_mix_flags"]).all()
	fl, cnt, first = c_oracle.multi_is_solved(k["solved_mix"])
	assert (fl == k["solved_mix_flags"]).all() and cnt == 3 and first == 3
	ch = orc.expand12(k["near_parents"])
	assert (orc.multi_is_solved(ch) == k["near_children_solved"]).all()
	assert k["near_children_solved"].sum() == 12
	_, fl = c_oracle.expand12(k["near_parents"])
	assert (fl.astype(bool) == k["near_children_solved"]).all()
	# truth tables of tests/test_cube.py:45-56
	s = orc.SOLVED.copy()
	for m, a in zip(((0, 1), (0, 0), (0, 1), (1, 1), (2, 0), (3, 0)), (False, True, False, False, False, False)):
		s = orc.rotate(s, *m)
		assert orc.is_solved(s) == a
	for m, a in zip(((3, 1), (2, 1), (1, 0), (0, 0)), (False, False, False, True)):
		s = orc.rotate(s, *m)
		assert orc.is_solved(s) == a


def test_sequence_scrambler_and_onehot(golden):
	k = golden["cube_kat"]
	np.random.seed(0)
	s, oh = orc.sequence_scrambler(4, 5, True)
	assert (s == k["k5_states"]).all() and (oh == k["k5_oh"]).all()
	assert sha(s) == "6063fc337784dd5b042a65a28396169baa7200224b35db36cdbe5ac1e638030e"
	assert sha(oh) == "d0c81699c5a07c852a657f5fcd91750aec7a4a047c7db9fae7e59803fbd3d85d"
	np.random.seed(0)
	s, _ = orc.sequence_scrambler(3, 4, False)
	assert (s == k["k5b_states"]).all()
	assert (orc.as_oh(k["k1_state"]) == k["oh_single"]).all() and orc.as_oh(k["k1_state"]).shape == (1, 480)
	assert (c_oracle.as_oh(k["k5_states"]) == k["k5_oh"]).all()
	assert (orc.as_oh(k["k5_states"]).sum(axis=1) == 20).all()


def test_actions_helpers(golden):
	text = golden["text"]
	assert orc.iter_actions(2).tolist() == text["iter_actions_2"] and orc.iter_actions(2).dtype == np.uint8
	assert orc.rev_actions(np.arange(12)).tolist() == text["rev_actions"]
	f, d = orc.indices_to_actions(np.arange(12))
	assert f.tolist() == [0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5] and d.tolist() == [1, 0] * 6   # tests/test_cube.py:123-127


def test_rendering(golden):
	k, text = golden["cube_kat"], golden["text"]
	s = orc.SOLVED.copy()
	assert orc.stringify(orc.as633(s)) == text["str_solved"]
	assert orc.stringify(orc.as633(orc.rotate(s, 0, 1))) == text["str_F"]
	for m in ((0, 0), (1, 0), (2, 0), (3, 0), (4, 0), (5, 0), (0, 1), (1, 1), (2, 1), (3, 1), (4, 1), (5, 1)):
		s = orc.rotate(s, *m)
	assert orc.stringify(orc.as633(s)) == text["str_all12"]
	for st, pic in zip(k["as633_states"], k["as633_out"]):
		assert (orc.as633(st) == pic).all()


def test_repr686(golden):
	k, text = golden["cube_kat"], golden["text"]
	s6 = k["r686_states"]
	assert (orc.multi_rotate686(s6, k["r686_faces"], k["r686_dirs"]) == k["r686_out"]).all()
	acts = (2 * k["r686_faces"] + (1 - k["r686_dirs"])).astype(np.uint8)
	assert (c_oracle.multi_rotate686(s6, acts) == k["r686_out"]).all()
	for i in range(8):
		for a in range(12):
			assert (orc.rotate686(s6[i], a // 2, 1 - a % 2) == k["r686_all12"][i, a]).all()
		assert (orc.as633_686(s6[i]) == k["r686_as633"][i]).all()
	assert (orc.as_correct686(s6) == k["r686_correct"]).all()
	st = orc.rotate686(orc.rotate686(orc.SOLVED686, 0, 1), 5, 0)
	assert (orc.as_correct686(st[None]) == k["r686_correct_FRp"]).all()
	# literal of tests/test_cube.py:158-165
	assert k["r686_correct_FRp"][0].tolist() == [
		[1, 1, 1, 1, -1, -1, -1, 1], [-1, 1, 1, 1, 1, 1, -1, -1], [-1, -1, -1, -1, -1, 1, 1, 1],
		[-1, -1, -1, -1, -1, 1, 1, 1], [-1, 1, 1, 1, 1, 1, -1, -1], [1, 1, -1, -1, -1, 1, 1, 1]]
	assert orc.stringify(orc.as633_686(orc.rotate686(orc.SOLVED686, 0, 1))) == text["str686_F"] == text["str_F"]
	assert (orc.as_oh686(s6).reshape(s6.shape) == s6).all()
