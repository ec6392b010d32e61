// Host-callable launchers of the cube kernels (definitions in rk_cube_kernels.hip).  Arguments are validated by
// the C-ABI layer (rk_api.hip); launchers assume non-null, aligned, in-range inputs.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "rk_tables.h"

namespace rk {

const Tables &host_tables();

void launch_expand12(const int8_t *parents, int8_t *children, uint8_t *solved, long long *stats, size_t n, hipStream_t st);
#ifdef RK_TUNING
void launch_expand12_variant(int variant, const int8_t *parents, int8_t *children, uint8_t *solved, long long *stats, size_t n,
                             int grid_blocks, hipStream_t st);
void launch_as_oh_variant(int tile, int grid_cap, const int8_t *states, void *out, int out_dtype, size_t n, hipStream_t st);
void tune_cell(void *device_cell_16_bytes);
void tune_pace_debug(void *device_buffer_32_bytes_per_tile_or_null);
#endif
void launch_expand12_soa(const uint32_t *parents, uint32_t *children, uint8_t *solved, long long *stats, size_t n, hipStream_t st);
void launch_states_soa(const int8_t *states, uint32_t *planes, size_t n, bool to_soa, hipStream_t st);
// with_flags: the goal test of the moved states in the same launch (flags nullable, stats nullable); action indices only
void launch_multi_rotate(const int8_t *states, const uint8_t *actions_or_faces, const uint8_t *dirs_or_null, int8_t *out,
                         size_t n, hipStream_t st, uint8_t *flags = nullptr, long long *stats = nullptr, bool with_flags = false);
void launch_multi_is_solved(const int8_t *states, uint8_t *flags, long long *stats, size_t n, hipStream_t st);
int read_bad_actions(hipStream_t st);
void set_pace_override(int mode);
int calibrate_pacing(bool force);
int pace_slot_of_device(int device);
void get_pacing(unsigned *tau_ps, int *source, float *us5);
void register_stream(hipStream_t st);
void forget_stream(hipStream_t st);
void launch_apply_sequences(const uint8_t *actions, int moves, int games, int with_solved, int only_last, int8_t *out,
                            hipStream_t st);
void launch_rollout_fanout(const uint8_t *actions, int moves, int games, int with_solved, int8_t *states, uint8_t *state_flags_or_null,
                           int8_t *children, uint8_t *child_flags, long long *stats_or_null, hipStream_t st);
void launch_as_oh(const int8_t *states, void *out, int out_dtype, size_t n, hipStream_t st);

void launch_rotate686(const int8_t *states, const uint8_t *actions, int8_t *out, size_t n_out, bool fanout, hipStream_t st,
                      uint8_t *flags = nullptr, long long *stats = nullptr);
void launch_is_solved686(const int8_t *states, uint8_t *flags, long long *stats, size_t n, hipStream_t st);
void launch_as_oh686(const int8_t *states, void *out, int out_dtype, size_t n, hipStream_t st);
void launch_as_correct686(const int8_t *states, float *out, size_t n, hipStream_t st);

}  // namespace rk
