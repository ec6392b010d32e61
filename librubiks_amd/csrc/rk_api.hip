// C ABI of librubiks_hip.so (declared in include/rubiks_hip.h): argument validation, error reporting and the
// host-pointer conveniences.  Nothing in here computes on the CPU: every entry either launches a HIP kernel or
// fails with an error code.
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <climits>

#include "../../include/rubiks_hip.h"
#include "rk_kernels.h"
#include "rk_error.h"

namespace rk {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof g_err, fmt, ap);
	va_end(ap);
	return code;
}

}  // namespace rk

using namespace rk;

namespace {

inline bool misaligned(const void *p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) != 0; }

inline int state_bytes(int repr) { return repr == RK_REPR_2024 ? STATE_BYTES : S686_BYTES; }

int check_repr(int repr)
{
	if (repr != RK_REPR_2024 && repr != RK_REPR_686) return fail(RK_EINVAL, "unknown representation %d", repr);
	return RK_OK;
}

// Scratch for the *_host entries: grow-only device buffers cached per host thread and device, so that a small call
// (one state through cube.rotate) costs two tiny copies and a launch instead of three hipMalloc/hipFree pairs.
struct ScratchSlot { void *p = nullptr; size_t cap = 0; int device = -1; };
thread_local ScratchSlot g_scratch[8];
thread_local int g_scratch_next = 0;

struct DevBuf {
	void *p = nullptr;
	int alloc(size_t bytes)
	{
		if (g_scratch_next >= 8) return fail(RK_ESTATE, "scratch slots exhausted");
		ScratchSlot &s = g_scratch[g_scratch_next++];
		int dev = 0;
		RK_HIP(hipGetDevice(&dev));
		if (bytes < 256) bytes = 256;
		if (s.device != dev || s.cap < bytes) {
			if (s.p && s.device == dev) (void)hipFree(s.p);
			s.p = nullptr; s.cap = 0; s.device = dev;
			RK_HIP(hipMalloc(&s.p, bytes));
			s.cap = bytes;
		}
		p = s.p;
		return RK_OK;
	}
};

// resets the slot cursor when a *_host entry starts (slots are handed out in call order, so sizes stay matched)
struct ScratchScope { ScratchScope() { g_scratch_next = 0; } };

// SMALL host calls go zero-copy: one page-locked, device-mapped buffer per host thread; the inputs are placed in it with a CPU
// memcpy, the kernel reads and writes it over the bus, the outputs are copied out after ONE stream synchronisation.  A staged
// call costs two or three hipMemcpyAsync round trips around its launch (29-37 us for one state through cube.rotate /
// is_solved, profiles/r04_reference_protocol.json -- what reference code that loops over single states sees); this way it
// costs the launch and the wait (10 000 states through multi_rotate: 410 KB, about 70 us from Python against 100 with a torch
// hop and staged copies).  Larger calls keep the staged copies (the CPU memcpy into and out of the buffer, and the kernel's
// accesses over the bus, then cost more than the copy engines do).
constexpr size_t ZERO_COPY_MAX = 1u << 20;            // staged bytes (inputs + outputs) up to which a call goes this way
struct PinnedBuf { char *host = nullptr; char *dev = nullptr; int device = -1; bool tried = false; };
thread_local PinnedBuf g_pinned;

inline size_t up256(size_t b) { return (b + 255) & ~(size_t)255; }

// the buffer of this thread for the current device, or nullptr (then the caller stages through device memory as before)
PinnedBuf *pinned_buffer()
{
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
	PinnedBuf &b = g_pinned;
	if (b.host && b.device == dev) return &b;
	if (b.tried && b.device == dev) return nullptr;
	if (b.host) { (void)hipHostFree(b.host); b.host = b.dev = nullptr; }
	b.device = dev; b.tried = true;
	void *h = nullptr, *d = nullptr;
	if (hipHostMalloc(&h, ZERO_COPY_MAX + 4096, hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
	if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipHostFree(h); return nullptr; }
	b.host = (char *)h; b.dev = (char *)d;
	return &b;
}

}  // namespace

extern "C" {

int rk_version(void) { return 100; }

const char *rk_last_error(void) { return g_err; }

int rk_init(int device)
{
	int count = 0;
	RK_HIP(hipGetDeviceCount(&count));
	if (device < 0 || device >= count) return fail(RK_EINVAL, "device %d out of range (%d visible)", device, count);
	RK_HIP(hipSetDevice(device));
	hipDeviceProp_t prop;
	RK_HIP(hipGetDeviceProperties(&prop, device));
	if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
		return fail(RK_EINVAL, "device %d is %s; this library only carries gfx950 code", device, prop.gcnArchName);
	(void)calibrate_pacing(false);          // once per device and process, about 4 ms; a calibration that cannot run keeps the compiled schedule
	return RK_OK;
}

int rk_tables(int repr, uint8_t *h_out)
{
	if (int e = check_repr(repr)) return e;
	if (!h_out) return fail(RK_EINVAL, "rk_tables: null output");
	const Tables &t = host_tables();
	if (repr == RK_REPR_2024) memcpy(h_out, t.lut, sizeof t.lut);
	else memcpy(h_out, t.perm686, sizeof t.perm686);
	return RK_OK;
}

int rk_face_definitions(uint8_t *h_out)
{
	if (!h_out) return fail(RK_EINVAL, "rk_face_definitions: null output");
	for (int f = 0; f < 6; f++) {
		uint8_t *o = h_out + 14 * f;
		for (int j = 0; j < 4; j++) {
			o[j] = FACES[f].corner[j];
			o[4 + j] = FACES[f].edge[j];
			o[10 + j] = NEIGHBOUR[f][j];
		}
		o[8] = FACES[f].fixed_ori;
		o[9] = FACES[f].flip;
	}
	return RK_OK;
}

int rk_solved(int repr, int8_t *h_out)
{
	if (int e = check_repr(repr)) return e;
	if (!h_out) return fail(RK_EINVAL, "rk_solved: null output");
	const Tables &t = host_tables();
	if (repr == RK_REPR_2024) memcpy(h_out, t.solved, STATE_BYTES);
	else memcpy(h_out, t.solved686, S686_BYTES);
	return RK_OK;
}

int rk_malloc(void **d_ptr, size_t bytes)
{
	if (!d_ptr) return fail(RK_EINVAL, "rk_malloc: null out pointer");
	RK_HIP(hipMalloc(d_ptr, bytes ? bytes : 16));
	return RK_OK;
}

int rk_free(void *d_ptr)
{
	RK_HIP(hipFree(d_ptr));
	return RK_OK;
}

int rk_memcpy_h2d(void *d_dst, const void *h_src, size_t bytes, void *stream)
{
	RK_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
	return RK_OK;
}

int rk_memcpy_d2h(void *h_dst, const void *d_src, size_t bytes, void *stream)
{
	RK_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
	return RK_OK;
}

int rk_memset(void *d_dst, int value, size_t bytes, void *stream)
{
	RK_HIP(hipMemsetAsync(d_dst, value, bytes, (hipStream_t)stream));
	return RK_OK;
}

int rk_stream_synchronize(void *stream)
{
	RK_HIP(hipStreamSynchronize((hipStream_t)stream));
	return RK_OK;
}

// ---- device-pointer entries ---------------------------------------------------------------------------------

int rk_multi_rotate(int repr, const int8_t *d_states, const uint8_t *d_actions, int8_t *d_out, size_t n, void *stream)
{
	if (int e = check_repr(repr)) return e;
	if (n == 0) return RK_OK;
	if (!d_states || !d_actions || !d_out) return fail(RK_EINVAL, "rk_multi_rotate: null pointer");
	if (misaligned(d_states, 4) || misaligned(d_out, 4)) return fail(RK_EINVAL, "rk_multi_rotate: state arrays must be 4-byte aligned");
	if (repr == RK_REPR_686 && (misaligned(d_out, 16) || misaligned(d_states, 16)))
		return fail(RK_EINVAL, "rk_multi_rotate: 6x8x6 state arrays must be 16-byte aligned");
	if (repr == RK_REPR_2024) launch_multi_rotate(d_states, d_actions, nullptr, d_out, n, (hipStream_t)stream);
	else {
		if (d_out == d_states) return fail(RK_EINVAL, "rk_multi_rotate: in-place not supported for the 6x8x6 representation");
		launch_rotate686(d_states, d_actions, d_out, n, false, (hipStream_t)stream);
	}
	RK_HIP(hipGetLastError());
	return RK_OK;
}

int rk_set_pacing(int mode)
{
	set_pace_override(mode);
	return RK_OK;
}

int rk_calibrate_pacing(int force)
{
	if (calibrate_pacing(force != 0) != 0) return fail(RK_EHIP, "rk_calibrate_pacing: no current device");
	return RK_OK;
}

int rk_get_pacing(unsigned int *tau_ps, int *source, float *h_us)
{
	get_pacing(tau_ps, source, h_us);
	return RK_OK;
}

int rk_pace_slot_of_device(int device) { return pace_slot_of_device(device); }

int rk_stream_register(void *stream)
{
	register_stream((hipStream_t)stream);
	return RK_OK;
}

int rk_stream_forget(void *stream)
{
	forget_stream((hipStream_t)stream);
	return RK_OK;
}

int rk_multi_rotate_solved(int repr, const int8_t *d_states, const uint8_t *d_actions, int8_t *d_out, uint8_t *d_flags, long long *d_stats,
                           size_t n, void *stream)
{
	if (int e = check_repr(repr)) return e;
	if (n == 0) return RK_OK;
	if (!d_states || !d_actions || !d_out) return fail(RK_EINVAL, "rk_multi_rotate_solved: null pointer");
	if (!d_flags && !d_stats) return fail(RK_EINVAL, "rk_multi_rotate_solved: flags or stats (or both) must be given");
	if (misaligned(d_states, 4) || misaligned(d_out, 4)) return fail(RK_EINVAL, "rk_multi_rotate_solved: state arrays must be 4-byte aligned");
	if (d_stats && misaligned(d_stats, 8)) return fail(RK_EINVAL, "rk_multi_rotate_solved: stats must be 8-byte aligned");
	if (repr == RK_REPR_2024) {
		launch_multi_rotate(d_states, d_actions, nullptr, d_out, n, (hipStream_t)stream, d_flags, d_stats, true);
	} else {
		// 6x8x6: the goal test reads the moved states back (two launches; the fused form exists for the 20-byte states)
		if (misaligned(d_out, 16) || misaligned(d_states, 16)) return fail(RK_EINVAL, "rk_multi_rotate_solved: 6x8x6 state arrays must be 16-byte aligned");
		if (d_out == d_states) return fail(RK_EINVAL, "rk_multi_rotate_solved: in-place not supported for the 6x8x6 representation");
		launch_rotate686(d_states, d_actions, d_out, n, false, (hipStream_t)stream);
		launch_is_solved686(d_out, d_flags, d_stats, n, (hipStream_t)stream);
	}
	RK_HIP(hipGetLastError());
	return RK_OK;
}

int rk_bad_actions_seen(int *h_seen, void *stream)
{
	if (!h_seen) return fail(RK_EINVAL, "rk_bad_actions_seen: null pointer");
	const int r = read_bad_actions((hipStream_t)stream);
	if (r < 0) return fail(RK_EHIP, "rk_bad_actions_seen: could not read the device flag");
	*h_seen = r;
	return RK_OK;
}

int rk_multi_rotate_fd(int repr, const int8_t *d_states, const uint8_t *d_faces, const uint8_t *d_dirs, int8_t *d_out,
                       size_t n, void *stream)
{
	if (int e = check_repr(repr)) return e;
	if (repr != RK_REPR_2024) return fail(RK_EINVAL, "rk_multi_rotate_fd: only the 20-byte representation; pass action indices for 6x8x6");
	if (n == 0) return RK_OK;
	if (!d_states || !d_faces || !d_dirs || !d_out) return fail(RK_EINVAL, "rk_multi_rotate_fd: null pointer");
	if (misaligned(d_states, 4) || misaligned(d_out, 4)) return fail(RK_EINVAL, "rk_multi_rotate_fd: state arrays must be 4-byte aligned");
	launch_multi_rotate(d_states, d_faces, d_dirs, d_out, n, (hipStream_t)stream);
	RK_HIP(hipGetLastError());
	return RK_OK;
}

int rk_expand12(int repr, const int8_t *d_parents, int8_t *d_children, uint8_t *d_solved, long long *d_stats, size_t n,
                void *stream)
{
	if (int e = check_repr(repr)) return e;
	if (n == 0) return RK_OK;
	if (!d_parents || !d_children) return fail(RK_EINVAL, "rk_expand12: null pointer");
	if (misaligned(d_parents, 4)) return fail(RK_EINVAL, "rk_expand12: parents must be 4-byte aligned");
	if (misaligned(d_children, 16)) return fail(RK_EINVAL, "rk_expand12: children must be 16-byte aligned");
	if (d_solved && misaligned(d_solved, 4)) return fail(RK_EINVAL, "rk_expand12: solved flags must be 4-byte aligned");
	if (d_stats && misaligned(d_stats, 8)) return fail(RK_EINVAL, "rk_expand12: stats must be 8-byte aligned");
	if (d_stats && !d_solved && repr == RK_REPR_2024) return fail(RK_EINVAL, "rk_expand12: stats need the solved-flag output");
	if (repr == RK_REPR_2024) {
		launch_expand12(d_parents, d_children, d_solved, d_stats, n, (hipStream_t)stream);
	} else {
		if (misaligned(d_parents, 16)) return fail(RK_EINVAL, "rk_expand12: 6x8x6 parents must be 16-byte aligned");
		launch_rotate686(d_parents, nullptr, d_children, 12 * n, true, (hipStream_t)stream, d_solved, d_stats);   // flags fused: one launch
	}
	RK_HIP(hipGetLastError());
	return RK_OK;
}

int rk_expand12_soa(const uint32_t *d_parents, uint32_t *d_children, uint8_t *d_solved, long long *d_stats, size_t n, void *stream)
{
	if (n == 0) return RK_OK;
	if (!d_parents || !d_children) return fail(RK_EINVAL, "rk_expand12_soa: null pointer");
	if (misaligned(d_parents, 4) || misaligned(d_children, 4)) return fail(RK_EINVAL, "rk_expand12_soa: planes must be 4-byte aligned");
	if (d_stats && (!d_solved || misaligned(d_stats, 8))) return fail(RK_EINVAL, "rk_expand12_soa: stats need flags and 8-byte alignment");
	launch_expand12_soa(d_parents, d_children, d_solved, d_stats, n, (hipStream_t)stream);
	RK_HIP(hipGetLastError());
	return RK_OK;
}

int rk_states_to_soa(const int8_t *d_states, uint32_t *d_planes, size_t n, void *stream)
{
	if (n == 0) return RK_OK;
	if (!d_states || !d_planes || misaligned(d_states, 4) || misaligned(d_planes, 4)) return fail(RK_EINVAL, "rk_states_to_soa: bad pointer");
	launch_states_soa(d_states, d_planes, n, true, (hipStream_t)stream);
	RK_HIP(hipGetLastError());
	return RK_OK;
}

int rk_states_from_soa(const uint32_t *d_planes, int8_t *d_states, size_t n, void *stream)
{
	if (n == 0) return RK_OK;
	if (!d_states || !d_planes || misaligned(d_states, 4) || misaligned(d_planes, 4)) return fail(RK_EINVAL, "rk_states_from_soa: bad pointer");
	launch_states_soa(d_states, const_cast<uint32_t *>(d_planes), n, false, (hipStream_t)stream);
	RK_HIP(hipGetLastError());
	return RK_OK;
}

#ifdef RK_TUNING
/* Tuning hook, deliberately outside the public header: other shapes of the fan-out kernel (benchmarks/tune_expand.py). */
int rkx_expand12_variant(int variant, const int8_t *d_parents, int8_t *d_children, uint8_t *d_solved, long long *d_stats, size_t n,
                         unsigned int *d_counter, int grid_blocks, void *stream)
{
	if (!d_parents || !d_children || !d_solved || misaligned(d_children, 16)) return fail(RK_EINVAL, "rkx_expand12_variant: bad argument");
	tune_cell(d_counter);
	launch_expand12_variant(variant, d_parents, d_children, d_solved, d_stats, n, grid_blocks, (hipStream_t)stream);
	RK_HIP(hipGetLastError());
	return RK_OK;
}

int rkx_pace_debug(void *d_buf) { tune_pace_debug(d_buf); return RK_OK; }

int rkx_as_oh_variant(int tile, int grid_cap, const int8_t *d_states, void *d_out, int out_dtype, size_t n, void *stream)
{
	if (!d_states || !d_out || misaligned(d_out, 16)) return fail(RK_EINVAL, "rkx_as_oh_variant: bad argument");
	launch_as_oh_variant(tile, grid_cap, d_states, d_out, out_dtype, n, (hipStream_t)stream);
	RK_HIP(hipGetLastError());
	return RK_OK;
}

#endif  /* RK_TUNING */

int rk_multi_is_solved(int repr, const int8_t *d_states, uint8_t *d_flags, long long *d_stats, size_t n, void *stream)
{
	if (int e = check_repr(repr)) return e;
	if (n == 0) return RK_OK;
	if (!d_states) return fail(RK_EINVAL, "rk_multi_is_solved: null states");
	if (misaligned(d_states, 4)) return fail(RK_EINVAL, "rk_multi_is_solved: states must be 4-byte aligned");
	if (d_stats && misaligned(d_stats, 8)) return fail(RK_EINVAL, "rk_multi_is_solved: stats must be 8-byte aligned");
	if (repr == RK_REPR_686 && misaligned(d_states, 16)) return fail(RK_EINVAL, "rk_multi_is_solved: 6x8x6 states must be 16-byte aligned");
	if (repr == RK_REPR_2024) launch_multi_is_solved(d_states, d_flags, d_stats, n, (hipStream_t)stream);
	else launch_is_solved686(d_states, d_flags, d_stats, n, (hipStream_t)stream);
	RK_HIP(hipGetLastError());
	return RK_OK;
}

int rk_apply_sequences(int repr, const uint8_t *d_actions, int depth, int games, int with_solved, int only_last,
                       int8_t *d_out, void *stream)
{
	if (int e = check_repr(repr)) return e;
	if (repr != RK_REPR_2024) return fail(RK_EINVAL, "rk_apply_sequences: only the 20-byte representation is implemented");
	if (depth < 0 || games < 0) return fail(RK_EINVAL, "rk_apply_sequences: negative size");
	with_solved = with_solved ? 1 : 0;
	const int moves = depth - with_solved;
	if (games == 0 || (moves < 0)) return RK_OK;
	if (!only_last && moves + with_solved == 0) return RK_OK;
	if (!d_out || (moves > 0 && !d_actions)) return fail(RK_EINVAL, "rk_apply_sequences: null pointer");
	if (misaligned(d_out, 4)) return fail(RK_EINVAL, "rk_apply_sequences: output must be 4-byte aligned");
	launch_apply_sequences(d_actions, moves, games, with_solved, only_last ? 1 : 0, d_out, (hipStream_t)stream);
	RK_HIP(hipGetLastError());
	return RK_OK;
}

int rk_rollout_fanout(int repr, const uint8_t *d_actions, int depth, int games, int with_solved, int8_t *d_states, uint8_t *d_state_flags,
                      int8_t *d_children, uint8_t *d_child_flags, long long *d_stats, void *stream)
{
	if (int e = check_repr(repr)) return e;
	if (repr != RK_REPR_2024) return fail(RK_EINVAL, "rk_rollout_fanout: only the 20-byte representation is implemented");
	if (depth < 0 || games < 0) return fail(RK_EINVAL, "rk_rollout_fanout: negative size");
	with_solved = with_solved ? 1 : 0;
	const int moves = depth - with_solved;
	if (games == 0 || moves < 0 || moves + with_solved == 0) return RK_OK;
	if (!d_states || !d_children || !d_child_flags || (moves > 0 && !d_actions)) return fail(RK_EINVAL, "rk_rollout_fanout: null pointer");
	if (misaligned(d_states, 4) || misaligned(d_children, 16) || misaligned(d_child_flags, 4))
		return fail(RK_EINVAL, "rk_rollout_fanout: states / child flags must be 4-byte aligned, children 16-byte aligned");
	if (d_stats && misaligned(d_stats, 8)) return fail(RK_EINVAL, "rk_rollout_fanout: stats must be 8-byte aligned");
	if ((size_t)games * depth > ((size_t)1 << 31)) return fail(RK_EINVAL, "rk_rollout_fanout: more than 2^31 states");
	launch_rollout_fanout(d_actions, moves, games, with_solved, d_states, d_state_flags, d_children, d_child_flags, d_stats, (hipStream_t)stream);
	RK_HIP(hipGetLastError());
	return RK_OK;
}

int rk_as_oh(int repr, const int8_t *d_states, void *d_out, int out_dtype, size_t n, void *stream)
{
	if (int e = check_repr(repr)) return e;
	if (out_dtype < RK_OH_F32 || out_dtype > RK_OH_BF16) return fail(RK_EINVAL, "rk_as_oh: unknown output dtype %d", out_dtype);
	if (n == 0) return RK_OK;
	if (!d_states || !d_out) return fail(RK_EINVAL, "rk_as_oh: null pointer");
	if (misaligned(d_states, 4)) return fail(RK_EINVAL, "rk_as_oh: states must be 4-byte aligned");
	if (misaligned(d_out, 16)) return fail(RK_EINVAL, "rk_as_oh: output must be 16-byte aligned");
	if (repr == RK_REPR_2024) launch_as_oh(d_states, d_out, out_dtype, n, (hipStream_t)stream);
	else launch_as_oh686(d_states, d_out, out_dtype, n, (hipStream_t)stream);
	RK_HIP(hipGetLastError());
	return RK_OK;
}

int rk_as_correct686(const int8_t *d_states, float *d_out, size_t n, void *stream)
{
	if (n == 0) return RK_OK;
	if (!d_states || !d_out) return fail(RK_EINVAL, "rk_as_correct686: null pointer");
	launch_as_correct686(d_states, d_out, n, (hipStream_t)stream);
	RK_HIP(hipGetLastError());
	return RK_OK;
}

// ---- host-pointer conveniences ------------------------------------------------------------------------------

static const long long STATS_INIT[2] = {0, LLONG_MAX};

int rk_multi_rotate_host(int repr, const int8_t *h_states, const uint8_t *h_actions, int8_t *h_out, size_t n, void *stream)
{
	if (int e = check_repr(repr)) return e;
	if (n == 0) return RK_OK;
	if (!h_states || !h_actions || !h_out) return fail(RK_EINVAL, "rk_multi_rotate_host: null pointer");
	for (size_t i = 0; i < n; i++)
		if (h_actions[i] >= N_ACTIONS) return fail(RK_EINVAL, "rk_multi_rotate_host: action %u at row %zu out of range", h_actions[i], i);
	const size_t sb = (size_t)state_bytes(repr);
	hipStream_t st = (hipStream_t)stream;
	if (up256(n * sb) * 2 + up256(n) <= ZERO_COPY_MAX) {
		if (PinnedBuf *b = pinned_buffer()) {                 // zero-copy: see ZERO_COPY_MAX
			const size_t o_act = up256(n * sb), o_out = o_act + up256(n);
			memcpy(b->host, h_states, n * sb);
			memcpy(b->host + o_act, h_actions, n);
			if (int e = rk_multi_rotate(repr, (const int8_t *)b->dev, (const uint8_t *)(b->dev + o_act), (int8_t *)(b->dev + o_out), n, stream)) return e;
			RK_HIP(hipStreamSynchronize(st));
			memcpy(h_out, b->host + o_out, n * sb);
			return RK_OK;
		}
	}
	ScratchScope scope;
	DevBuf in, act, out;
	if (int e = in.alloc(n * sb)) return e;
	if (int e = act.alloc(n)) return e;
	if (int e = out.alloc(n * sb)) return e;
	RK_HIP(hipMemcpyAsync(in.p, h_states, n * sb, hipMemcpyHostToDevice, st));
	RK_HIP(hipMemcpyAsync(act.p, h_actions, n, hipMemcpyHostToDevice, st));
	if (int e = rk_multi_rotate(repr, (const int8_t *)in.p, (const uint8_t *)act.p, (int8_t *)out.p, n, stream)) return e;
	RK_HIP(hipMemcpyAsync(h_out, out.p, n * sb, hipMemcpyDeviceToHost, st));
	RK_HIP(hipStreamSynchronize(st));
	return RK_OK;
}

int rk_expand12_host(int repr, const int8_t *h_parents, int8_t *h_children, uint8_t *h_solved, long long *h_stats, size_t n,
                     void *stream)
{
	if (int e = check_repr(repr)) return e;
	if (h_stats) { h_stats[0] = 0; h_stats[1] = -1; }
	if (n == 0) return RK_OK;
	if (!h_parents || !h_children) return fail(RK_EINVAL, "rk_expand12_host: null pointer");
	const size_t sb = (size_t)state_bytes(repr);
	hipStream_t st = (hipStream_t)stream;
	if (!h_stats && up256(n * sb) + up256(12 * n * sb) + up256(12 * n) <= ZERO_COPY_MAX) {     // (the counters are atomics: they stay in device memory)
		if (PinnedBuf *b = pinned_buffer()) {                 // zero-copy: see ZERO_COPY_MAX
			const size_t o_ch = up256(n * sb), o_fl = o_ch + up256(12 * n * sb);
			memcpy(b->host, h_parents, n * sb);
			if (int e = rk_expand12(repr, (const int8_t *)b->dev, (int8_t *)(b->dev + o_ch), h_solved ? (uint8_t *)(b->dev + o_fl) : nullptr, nullptr, n, stream)) return e;
			RK_HIP(hipStreamSynchronize(st));
			memcpy(h_children, b->host + o_ch, 12 * n * sb);
			if (h_solved) memcpy(h_solved, b->host + o_fl, 12 * n);
			return RK_OK;
		}
	}
	ScratchScope scope;
	DevBuf in, ch, fl, stt;
	if (int e = in.alloc(n * sb)) return e;
	if (int e = ch.alloc(12 * n * sb)) return e;
	if (int e = fl.alloc(12 * n)) return e;
	if (int e = stt.alloc(sizeof STATS_INIT)) return e;
	RK_HIP(hipMemcpyAsync(in.p, h_parents, n * sb, hipMemcpyHostToDevice, st));
	RK_HIP(hipMemcpyAsync(stt.p, STATS_INIT, sizeof STATS_INIT, hipMemcpyHostToDevice, st));
	if (int e = rk_expand12(repr, (const int8_t *)in.p, (int8_t *)ch.p, (uint8_t *)fl.p, (long long *)stt.p, n, stream)) return e;
	RK_HIP(hipMemcpyAsync(h_children, ch.p, 12 * n * sb, hipMemcpyDeviceToHost, st));
	if (h_solved) RK_HIP(hipMemcpyAsync(h_solved, fl.p, 12 * n, hipMemcpyDeviceToHost, st));
	long long stats[2];
	RK_HIP(hipMemcpyAsync(stats, stt.p, sizeof stats, hipMemcpyDeviceToHost, st));
	RK_HIP(hipStreamSynchronize(st));
	if (h_stats) { h_stats[0] = stats[0]; h_stats[1] = stats[0] ? stats[1] : -1; }
	return RK_OK;
}

int rk_multi_is_solved_host(int repr, const int8_t *h_states, uint8_t *h_flags, long long *h_stats, size_t n, void *stream)
{
	if (int e = check_repr(repr)) return e;
	if (h_stats) { h_stats[0] = 0; h_stats[1] = -1; }
	if (n == 0) return RK_OK;
	if (!h_states) return fail(RK_EINVAL, "rk_multi_is_solved_host: null pointer");
	const size_t sb = (size_t)state_bytes(repr);
	hipStream_t st = (hipStream_t)stream;
	if (!h_stats && up256(n * sb) + up256(n) <= ZERO_COPY_MAX) {     // (the counters are atomics: they stay in device memory)
		if (PinnedBuf *b = pinned_buffer()) {                 // zero-copy: see ZERO_COPY_MAX
			const size_t o_fl = up256(n * sb);
			memcpy(b->host, h_states, n * sb);
			if (int e = rk_multi_is_solved(repr, (const int8_t *)b->dev, (uint8_t *)(b->dev + o_fl), nullptr, n, stream)) return e;
			RK_HIP(hipStreamSynchronize(st));
			if (h_flags) memcpy(h_flags, b->host + o_fl, n);
			return RK_OK;
		}
	}
	ScratchScope scope;
	DevBuf in, fl, stt;
	if (int e = in.alloc(n * sb)) return e;
	if (int e = fl.alloc(n)) return e;
	RK_HIP(hipMemcpyAsync(in.p, h_states, n * sb, hipMemcpyHostToDevice, st));
	if (h_stats) {                                   // the counters cost two extra copies: only when asked for
		if (int e = stt.alloc(sizeof STATS_INIT)) return e;
		RK_HIP(hipMemcpyAsync(stt.p, STATS_INIT, sizeof STATS_INIT, hipMemcpyHostToDevice, st));
	}
	if (int e = rk_multi_is_solved(repr, (const int8_t *)in.p, (uint8_t *)fl.p, (long long *)stt.p, n, stream)) return e;
	if (h_flags) RK_HIP(hipMemcpyAsync(h_flags, fl.p, n, hipMemcpyDeviceToHost, st));
	long long stats[2] = {0, 0};
	if (h_stats) RK_HIP(hipMemcpyAsync(stats, stt.p, sizeof stats, hipMemcpyDeviceToHost, st));
	RK_HIP(hipStreamSynchronize(st));
	if (h_stats) { h_stats[0] = stats[0]; h_stats[1] = stats[0] ? stats[1] : -1; }
	return RK_OK;
}

int rk_apply_sequences_host(int repr, const uint8_t *h_actions, int depth, int games, int with_solved, int only_last,
                            int8_t *h_out, void *stream)
{
	if (int e = check_repr(repr)) return e;
	if (repr != RK_REPR_2024) return fail(RK_EINVAL, "rk_apply_sequences_host: only the 20-byte representation is implemented");
	if (depth < 0 || games < 0) return fail(RK_EINVAL, "rk_apply_sequences_host: negative size");
	with_solved = with_solved ? 1 : 0;
	const int moves = depth - with_solved;
	if (games == 0 || moves < 0) return RK_OK;
	const size_t rows = only_last ? 1 : (size_t)(moves + with_solved);
	if (rows == 0) return RK_OK;
	if (!h_out || (moves > 0 && !h_actions)) return fail(RK_EINVAL, "rk_apply_sequences_host: null pointer");
	const size_t nact = (size_t)moves * games;
	for (size_t i = 0; i < nact; i++)
		if (h_actions[i] >= N_ACTIONS) return fail(RK_EINVAL, "rk_apply_sequences_host: action %u out of range", h_actions[i]);
	hipStream_t st = (hipStream_t)stream;
	const size_t out_bytes = (size_t)games * rows * STATE_BYTES;
	if (up256(nact) + up256(out_bytes) <= ZERO_COPY_MAX) {
		if (PinnedBuf *b = pinned_buffer()) {                 // zero-copy: see ZERO_COPY_MAX
			const size_t o_out = up256(nact);
			if (nact) memcpy(b->host, h_actions, nact);
			if (int e = rk_apply_sequences(repr, (const uint8_t *)b->dev, depth, games, with_solved, only_last, (int8_t *)(b->dev + o_out), stream)) return e;
			RK_HIP(hipStreamSynchronize(st));
			memcpy(h_out, b->host + o_out, out_bytes);
			return RK_OK;
		}
	}
	ScratchScope scope;
	DevBuf act, out;
	if (int e = act.alloc(nact)) return e;
	if (int e = out.alloc((size_t)games * rows * STATE_BYTES)) return e;
	if (nact) RK_HIP(hipMemcpyAsync(act.p, h_actions, nact, hipMemcpyHostToDevice, st));
	if (int e = rk_apply_sequences(repr, (const uint8_t *)act.p, depth, games, with_solved, only_last, (int8_t *)out.p, stream)) return e;
	RK_HIP(hipMemcpyAsync(h_out, out.p, (size_t)games * rows * STATE_BYTES, hipMemcpyDeviceToHost, st));
	RK_HIP(hipStreamSynchronize(st));
	return RK_OK;
}

int rk_as_oh_host(int repr, const int8_t *h_states, void *d_out, int out_dtype, size_t n, void *stream)
{
	if (int e = check_repr(repr)) return e;
	if (n == 0) return RK_OK;
	if (!h_states || !d_out) return fail(RK_EINVAL, "rk_as_oh_host: null pointer");
	const size_t sb = (size_t)state_bytes(repr);
	hipStream_t st = (hipStream_t)stream;
	if (up256(n * sb) <= ZERO_COPY_MAX) {
		if (PinnedBuf *b = pinned_buffer()) {                 // zero-copy: see ZERO_COPY_MAX
			memcpy(b->host, h_states, n * sb);
			if (int e = rk_as_oh(repr, (const int8_t *)b->dev, d_out, out_dtype, n, stream)) return e;
			RK_HIP(hipStreamSynchronize(st));                 // the buffer belongs to the next call from here on
			return RK_OK;
		}
	}
	ScratchScope scope;
	DevBuf in;
	if (int e = in.alloc(n * sb)) return e;
	RK_HIP(hipMemcpyAsync(in.p, h_states, n * sb, hipMemcpyHostToDevice, st));
	if (int e = rk_as_oh(repr, (const int8_t *)in.p, d_out, out_dtype, n, stream)) return e;
	RK_HIP(hipStreamSynchronize(st));
	return RK_OK;
}

}  // extern "C"
