/* libunreal_hip.so -- C ABI of the MI355X (gfx950) UNREAL actor-learner hot path.
 *
 * The reference (kvas7andy/unreal, /root/reference) has no FFI of its own: the path sits behind a
 * Python class surface and, below it, TensorFlow's op registry.  Every entry point here replaces one
 * reference call site (cited per function, file:line under /root/reference) and is what a binding
 * for that call would bind.  INTEGRATION.md shows the ctypes stub.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes.  ALL pointers are DEVICE pointers owned by the caller
 *     (the Python host allocates them through PyTorch-ROCm); nothing is allocated or freed here.
 *   - `stream` is a hipStream_t passed as void*; every call is asynchronous on that stream.
 *   - return 0 = OK, -22 = invalid argument (nothing launched), -5 = launch failure.  Never throws.
 *   - No global state; not thread-safe per buffer (one host thread per GPU by contract).
 *   - Frames are uint8 NHWC 84x84x3 (21,168 B); a "frame index" f addresses frames + f*21168.
 *   - Ring: actor b owns H1 = H+1 slots; absolute frame i lives in slot i % H1; per-slot metadata
 *     arrays are [B][H1]; the current observation of actor b is slot count[b] % H1.
 */
#ifndef UNREAL_HIP_H
#define UNREAL_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define UNREAL_FRAME_BYTES 21168
#define UNREAL_PC_CELLS 400
/* unreal_gemm_f32 flags */
#define UNREAL_GEMM_RELU 1
#define UNREAL_GEMM_ACCUM 2
#define UNREAL_GEMM_ATOMIC 4
#define UNREAL_GEMM_RELU_MASK 8
#define UNREAL_GEMM_RELU_BITS 16   /* split_nt only: mask = uint16 bit words (unreal_encoder_fwd relu_bits), ldm in words */

/* ---- environment (environment/maze_environment.py:50-55,98-128; environment/environment.py:88-102;
 *      train/experience.py:63-93 add_frame; train/trainer.py:194-205,264-296 reset rules) ---------- */
int unreal_maze_step(int B, int H1, const int* actions, const int* active, int* pos, int* last_action,
                     float* last_reward, int* count, uint8_t* frames, float* r_reward, int* r_action,
                     int* r_terminal, int* r_last_action, float* r_last_reward, float* r_pc,
                     float* out_reward, int* out_terminal, float* episode_reward, float* score_out,
                     int* score_valid, int reset_on_terminal, int track_score, void* stream);
/* the same step with the rollout loop's bookkeeping (train/trainer.py:236-296) in the same launch: `active` is read
 * AND updated (an actor leaves the rollout at its terminal, the reference's `break`), active_log_t / n_steps /
 * terminal_end as unreal_rollout_advance writes them, next_idx = ring index of every actor's next observation
 * (unreal_ring_cur_idx) and next_lar = the [one-hot last action | last reward] columns of the next step's LSTM-input
 * rows (unreal_lar_fill): three ~5 us launches per rollout step less.  reset_on_terminal = track_score = 1. */
int unreal_maze_rollout_step(int B, int H1, const int* actions, int* pos, int* last_action, float* last_reward, int* count,
                             uint8_t* frames, float* r_reward, int* r_action, int* r_terminal, int* r_last_action,
                             float* r_last_reward, float* r_pc, float* out_reward, int* out_terminal,
                             float* episode_reward, float* score_out, int* score_valid, int* active,
                             int* active_log_t, int* n_steps, int* terminal_end, int* next_idx /*nullable*/,
                             float* next_lar /*nullable*/, int lar_ld, int lar_col0, int A,
                             int idx_base_actor /* next_idx[b] = (idx_base_actor + b) * H1 + slot: a half-batch whose
                                                   rows index the whole ring */, void* stream);
/* unreal_policy_step + unreal_maze_rollout_step in ONE launch (trainer.py:236-296: run_base_policy_and_value, choose_action,
 * environment.process of one rollout step): the workgroup that steps an actor first computes its pi / V from the feature row
 * X[b] (K = 256) and draws its action from u[b] -- bit-identical to the two-launch path.  A must be 4 (the maze). */
int unreal_maze_policy_rollout_step(int B, int H1, const float* X, int ldx, const float* Wp, const float* bp, const float* Wv,
                                    const float* bv, const double* u, float* pi_out, float* v_out, int* actions_out, int* pos,
                                    int* last_action, float* last_reward, int* count, uint8_t* frames, float* r_reward,
                                    int* r_action, int* r_terminal, int* r_last_action, float* r_last_reward, float* r_pc,
                                    float* out_reward, int* out_terminal, float* episode_reward, float* score_out,
                                    int* score_valid, int* active, int* active_log_t, int* n_steps, int* terminal_end,
                                    int* next_idx /*nullable*/, float* next_lar /*nullable*/, int lar_ld, int lar_col0, int A,
                                    int idx_base_actor, void* stream);
int unreal_maze_reset(int B, int H1, const int* mask, int* pos, int* last_action, float* last_reward,
                      const int* count, uint8_t* frames, void* stream);
/* host-fed environments (environment/lab_environment.py:78-119 contract): `staged` holds one uint8
 * frame per actor (post-reset observation where terminals[b] != 0) */
int unreal_hostfed_step(int B, int H1, const uint8_t* staged, const int* actions, const float* rewards,
                        const int* terminals, const int* active, int* last_action, float* last_reward, int* count,
                        uint8_t* frames, float* r_reward, int* r_action, int* r_terminal, int* r_last_action,
                        float* r_last_reward, float* r_pc, float* out_reward, int* out_terminal,
                        float* episode_reward, float* score_out, int* score_valid, int reset_on_terminal,
                        int track_score, int clip_reward, float pc_denom, void* stream);
int unreal_hostfed_reset(int B, int H1, const int* mask, const uint8_t* staged, int* last_action, float* last_reward,
                         const int* count, uint8_t* frames, void* stream);
/* generic _calc_pixel_change on stored uint8 frames: out[n][400] = sum_{4x4x3}|new-old| / denom */
int unreal_pixel_change_u8(int N, const uint8_t* frames, const int* idx_new, const int* idx_old,
                           float denom, float* out, void* stream);

/* ---- counter RNG (stands in for the shared numpy RandomState of main.py:213) ----------
 * out[i] = draw number (i / row_len) * row_stride + col0 + i % row_len of Philox stream (seed, stream_id): with
 * row_len = actors of this rank, row_stride = actors of all ranks, col0 = first actor of this rank, a sharded job
 * draws exactly what one process holding every actor would (row_len = row_stride = n, col0 = 0: a plain stream). */
int unreal_philox_uniform(uint64_t seed, uint64_t stream_id, int n, int row_len, int row_stride, int col0,
                          double* out, void* stream);
int unreal_philox_randint(uint64_t seed, uint64_t stream_id, int n, int row_len, int row_stride, int col0, int high,
                          int* out, void* stream);

/* ---- replay sampling (train/experience.py:100-118, 121-153; train/trainer.py:427-434) ------------ */
int unreal_replay_sample_seq(int B, int H, int H1, int L, const int* start_draw, const int* count,
                             const int* r_terminal, int* seq_idx /*[L][B]*/, int* seq_len /*[B]*/, void* stream);
/* mode 0: this fork's buckets (reward > 0 | rest); mode 1: upstream / Lab replay (reward != 0 | reward == 0,
 * train/experience_lab_ver.py:76-80, 124-141) */
int unreal_replay_sample_rp(int B, int H, int H1, const int* coin, const double* u, const int* count,
                            const float* r_reward, int* rp_idx /*[B][3]*/, int* rp_class /*[B]*/, int mode,
                            void* stream);

/* ---- return scans (train/trainer.py:298-324, 354-372, 394-406), fp64 like the reference ----------- */
int unreal_base_returns(int B, int T, const float* rewards, const float* values, const int* n_steps,
                        const float* boot_v, const int* terminal_end, double gamma, float* R_out, float* adv_out,
                        void* stream);
int unreal_vr_returns(int B, int L, const int* seq_idx, const int* seq_len, const float* r_reward,
                      const int* r_terminal, const float* boot_v, double gamma, float* R_out, void* stream);
int unreal_pc_returns(int B, int L, const int* seq_idx, const int* seq_len, const float* r_pc,
                      const int* r_terminal, const float* boot_qmax, double gamma_pc, float* R_out, void* stream);

/* ---- rollout bookkeeping (train/experience.py:35-46; train/trainer.py:236-296; model.py:625-628) -- */
/* clip_reward != 0: the reward column is np.clip(r, -1, 1) (ExperienceFrame of train/experience_lab_ver.py:14,18) */
int unreal_lar_fill(int rows, int A, const int* last_action, const float* last_reward, const int* idx, float* xcat,
                    int ld, int col0, int clip_reward, void* stream);
/* objective vectors of multimodal environments (environment/indoor_environment.py:70-73,113; train/experience.py:42-44;
 * model/model.py:144,343): one [obj] fp32 row per ring slot next to the frame.  put: staged[b] -> the current slot
 * of every active actor.  fill: xcat[row][col0..col0+obj) = objective of frame idx[row] shifted by slot_offset slots
 * inside its actor's ring (-1 for the bootstrap value of train/trainer.py:300, which is fed the objective of the
 * previous frame's state). */
int unreal_objective_put(int B, int H1, int obj, const int* count, const int* active, const float* staged,
                         float* r_objective, void* stream);
int unreal_objective_fill(int rows, int obj, int H1, const float* r_objective, const int* idx, int slot_offset,
                          float* xcat, int ld, int col0, void* stream);
int unreal_gather_i32(int rows, const int* src, const int* idx, int* out, void* stream);
int unreal_rollout_advance(int B, const int* terminal_t, int* active, int* active_log_t, int* n_steps,
                           int* terminal_end, void* stream);
int unreal_seq_mask(int B, int T, const int* seq_len, int* mask, void* stream);
int unreal_reset_state(int B, const int* terminal_end, float* c, float* h, void* stream);
/* out[b] = (b0 + b) * H1 + count[b] % H1: with `count` pointing at actor b0 of a larger ring, indices into THAT ring */
int unreal_ring_cur_idx(int B, int H1, int b0, const int* count, int* out /*[B]*/, void* stream);
int unreal_seq_last_idx(int B, const int* seq_idx, const int* seq_len, int* out /*[B]*/, void* stream);
/* stats[3] (double) += {env steps, finished episodes, sum of their scores}; clears score_valid
 * (train/trainer.py:635-636 return value) */
int unreal_rollout_stats(int B, const int* n_steps, int* score_valid, const float* score_out, double* stats,
                         void* stream);

/* ---- conv encoder (model/model.py:281-289,786-787) and its gradient ------------------------------ */
int unreal_encoder_fwd(int N, const uint8_t* frames, const int* frame_idx, float frame_scale, const float* W1,
                       const float* b1, const float* W2, const float* b2, float* c1_out /*nullable [N][400][16]*/,
                       float* f2_out /*[N][2592]*/,
                       uint16_t* relu_bits /*nullable [N][81][2]: bit c of word [n][pos][h] = f2[n][pos][16h + c] > 0,
                                             i.e. bit (j % 16) of word j / 16 of row n; UNREAL_GEMM_RELU_BITS reads it */,
                       float* f2_absmax /*nullable absmax slot: max of f2_out, see unreal_absmax_f32*/,
                       float* c1_absmax /*nullable absmax slot: max of the conv1 activation (c1_out)*/,
                       const void* prepared /*nullable: block written by unreal_encoder_prepare for THESE W1, b1, W2 and
                                              frame_scale; without it every workgroup derives scales and operand
                                              fragments itself (identical results)*/,
                       void* stream);
/* The weights' share of unreal_encoder_fwd's prologue, once per weight update instead of once per workgroup and launch:
 * power-of-two scales of W1 / W2 / the conv1 planes and the fp16 hi + lo MFMA operand fragments of both convolutions ->
 * `prepared` (UNREAL_ENCODER_PREPARED_BYTES, 16-byte aligned). */
#define UNREAL_ENCODER_PREPARED_BYTES 45072
int unreal_encoder_prepare(const float* W1, const float* b1, const float* W2, float frame_scale, void* prepared,
                           long prepared_bytes, void* stream);
/* c1_absmax / d2_absmax: absmax slots covering c1_saved / d2 (the kernel keeps both as fp16 hi + lo planes with one
 * power-of-two scale per tensor, like the split GEMMs; the round-2 operand format needed none). */
int unreal_encoder_bwd(int N, const uint8_t* frames, const int* frame_idx, float frame_scale, const float* W2,
                       const float* c1_saved, const float* c1_absmax, const float* d2, const float* d2_absmax, float* dW1,
                       float* db1, float* dW2, float* db2, void* stream);

/* ---- dense layers: tf.matmul call sites model/model.py:314,334,423 and BasicLSTMCell 110,346-351 -- */
int unreal_gemm_f32(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B,
                    int ldb, float* C, int ldc, const float* bias, const float* mask, int ldm, int flags,
                    int splitk, void* stream);
/* "absmax slot" = one float in device memory holding max |x| over a tensor (0-initialised by the caller; producers
 * max their outputs into it atomically).  The fp16x2 GEMMs below derive each operand's power-of-two scale from it, so
 * the slot handed to a GEMM must cover every element the GEMM reads (a larger value only costs precision).
 * unreal_absmax_f32 is the stand-alone reduction: slot = max(slot, max |x[r][c]|). */
int unreal_absmax_f32(int rows, int cols, const float* x, int ld, float* slot, void* stream);
/* C = A[M,K] * W[N,K]^T with fp32-grade error on the 16-bit matrix cores (csrc/gemm_split.hip, round 3): each operand
 * is x * 2^k = hi + lo with hi, lo fp16 (k per TENSOR, from its absmax slot), the three term pairs hh, hl, lh are
 * accumulated in fp32 on v_mfma_f32_32x32x16_f16 and un-scaled exactly.  A is fp32; W2 is the weight matrix as a
 * pre-split shadow made by unreal_split_f16x2 with the SAME w_absmax slot: plane t (0 = hi, 1 = lo) at
 * W2 + t*plane_stride, row n at + n*ldw (ldw a multiple of 8 and >= K rounded up to 32, padding zero).  c_absmax
 * (nullable): receives max |C| of what this call stores; refused (-22) together with UNREAL_GEMM_ATOMIC / splitk > 1,
 * whose epilogue adds partial tiles and never sees a finished element.  Same epilogue flags as unreal_gemm_f32.  Used
 * for the forward and dgrad GEMMs of the dense layers (tf.matmul call sites model/model.py:314,334,423). */
int unreal_gemm_f32_split_nt(int M, int N, int K, const float* A, int lda, const float* a_absmax, const uint16_t* W2,
                             int ldw, long plane_stride, const float* w_absmax, float* C, int ldc, float* c_absmax,
                             const float* bias,
                             const void* mask /* fp32 [M][ldm] (RELU_MASK: keep where > 0) or uint16 bit words [M][ldm]
                                                 (RELU_BITS: keep column j where bit j % 16 of word j / 16 is set) */,
                             int ldm, int flags, int splitk, void* stream);
/* The same product for FEW ROWS and a LONG K (the fc 2592 -> 256 of a rollout step at <= 1024 rows): the K range runs as
 * `splitk` >= 2 slabs in separate workgroups, their partial products go to `partials` (>= splitk * M * pad4(N) floats,
 * 16-byte aligned), a second launch adds them in slab order (deterministic), applies bias / ReLU (flags: 0 or 1) and
 * commits max |C| to c_absmax (nullable). */
int unreal_gemm_f32_split_nt_slabs(int M, int N, int K, const float* A, int lda, const float* a_absmax, const uint16_t* W3,
                                   int ldw, long plane_stride, const float* w_absmax, float* C, int ldc, float* c_absmax,
                                   const float* bias, int flags, int splitk, float* partials, long partial_floats,
                                   void* stream);
/* splitk > 1 needs UNREAL_GEMM_ATOMIC (K slabs added into a pre-initialised C with fp32 atomics; no ReLU / mask /
 * ACCUM then).  Measured at the per-step shapes (4096 rows): 4096x256x2592 61 -> 50 us at splitk 4, 4096x256x1024
 * slower (the atomics cost what the extra workgroups buy), so the trainer keeps splitk = 1 there. */
/* wgrad on the same scheme: C[M,N] += A[K,M]^T * B[K,N] (k = the row index of both activations), split-K with fp32
 * atomics into the caller's (pre-initialised) C.  Both operands are split while they are transposed into LDS.
 * lda, ldb multiples of 4 and A, B 16-byte aligned (else -22: use unreal_gemm_f32 transA=1).
 * colsum (nullable): colsum[n] += sum_k B[k][n], the bias gradient that belongs to this weight gradient, summed
 * from the B tiles the kernel stages anyway (saves a separate pass over B). */
/* a_absmax / b_absmax: absmax slots covering A and B (both are split as fp16 hi + lo like the NT operands; the MFMA
 * accumulators are flushed into a second fp32 set every 8 K tiles, which keeps the error under the fp32-MFMA kernel's at
 * K = 81,920). */
int unreal_gemm_f32_split_tn(int M, int N, int K, const float* A, int lda, const float* a_absmax, const float* B, int ldb,
                             const float* b_absmax, float* C, int ldc, float* colsum, int splitk, void* stream);
/* bf16x3 shadow of a weight matrix src[rows][cols]: dst[t][r][c] (transpose = 0) or dst[t][c][r] (transpose = 1),
 * t = 0..2 the bf16 terms (sum of the three == src to 2^-24 relative).  dst padding is left untouched (zero it once).
 * (The round-2 operand format; the trainer's shadows are fp16x2 now.) */
int unreal_split_bf16x3(int rows, int cols, const float* src, int ld_src, int transpose, int row_perm, uint16_t* dst,
                        int ld_dst, long plane_stride, void* stream);
/* fp16x2 shadow: dst[0] = hi, dst[1] = lo of src * 2^k, k from w_absmax (a slot that already holds max |src| over the
 * WHOLE matrix the consuming GEMM multiplies by).  Same layouts / row_perm as above.  Refreshed after every RMSProp step /
 * checkpoint restore. */
/* Every weight shadow of a network in one pass (round 4): `abs_descs` = n_abs records {const float* src; float* wmax; long n;
 * long block0} (contiguous matrices; block0 = first block of the record in a grid of 8192-float blocks, ascending), `split_descs`
 * = n_split records {const float* src; uint16_t* dst; const float* wmax; long rows, cols, ld_src, transpose, row_perm, ld_dst,
 * plane, tiles_x, block0} (32 x 32 tiles; the arguments of unreal_split_f16x2).  The wmax slots must be zeroed by the caller;
 * element for element the arithmetic of unreal_absmax_f32 + unreal_split_f16x2 (identical bits), in two launches instead of
 * three per matrix.  Both tables are device memory. */
int unreal_shadow_refresh_multi(const void* abs_descs, int n_abs, int abs_blocks, const void* split_descs, int n_split,
                                int split_blocks, void* stream);
int unreal_split_f16x2(int rows, int cols, const float* src, int ld_src, int transpose, int row_perm, uint16_t* dst,
                       int ld_dst, long plane_stride, const float* w_absmax, void* stream);
/* row_perm = 1 (1024 output rows only): LSTM gate interleave, output row of column n = g*256 + u of the kernel is
 * (u/16)*64 + g*16 + u%16, the layout unreal_lstm_step_fwd multiplies by. */
/* One BasicLSTMCell step (model/model.py:110,346-351; gates i,j,f,o, forget_bias 1) on the split-operand path with the
 * gate math in the GEMM epilogue.  gates [rows][1024]: out = activated gates (saved for the backward).
 *   x == NULL: gates holds the input-half pre-activations (x * Wx, hoisted over all T steps of a training sequence)
 *              on entry; W3 = gate-interleaved bf16x3 shadow of the kernel's recurrent rows [1024][256]
 *              (unreal_split_bf16x3 transpose = 1, row_perm = 1) and only h_prev[rows,256] * Wh is multiplied here;
 *   x != NULL: the cell's own product [x | h_prev] @ kernel in ONE launch (a rollout step, where the input half cannot
 *              be hoisted): x[rows][Kx] (row stride ldx), W3 = gate-interleaved shadow of the WHOLE kernel,
 *              [1024][pad32(Kx) + 256]: columns [0, Kx) the input rows, zeros up to pad32(Kx), then the recurrent rows. */
/* x_absmax: slot covering x[rows][Kx] (required when x != NULL; |h_prev| < 1 is covered by the kernel itself);
 * w_absmax: the slot the shadow was made with. */
int unreal_lstm_step_fwd(int rows, const float* x, int ldx, int Kx, const float* x_absmax, const float* h_prev, int ld_hprev,
                         const uint16_t* W2, int ldw, long plane_stride, const float* w_absmax, float* gates,
                         const float* bias, const float* c_prev, float* c_out, float* h_out, int ld_h, void* stream);
/* BPTT through one recurrence step, fused: dh_rec = d_gates[rows,1024] (step t) * Wh^T on the split-operand path
 * (Wh3 = natural-layout shadow of the kernel's recurrent rows, [256][1024]), and in the same launch the gate backward
 * of step t-1 with dh = dh_above + dh_rec: dpre (d_gates of step t-1), dc_io in/out.  Same arithmetic and order as
 * unreal_gemm_f32_split_nt followed by unreal_lstm_gates_bwd (bit-identical), without materialising dh_rec. */
/* a_absmax: slot covering d_gates; dpre_absmax0 / 1 (nullable): receive max |dpre| (the next step's a_absmax, and the
 * slot of the whole sequence that the fc dgrad reads). */
int unreal_lstm_bptt_step(int rows, const float* d_gates, const float* a_absmax, const uint16_t* Wh2, int ldw,
                          long plane_stride, const float* w_absmax, const float* dh_above, float* dc_io,
                          const float* gates_act, const float* c_prev, const float* c_new, float* dpre,
                          float* dpre_absmax0, float* dpre_absmax1, void* stream);
int unreal_lstm_gates_fwd(int rows, const float* pre, const float* bias, const float* c_prev, float* gates_act,
                          float* c_out, float* h_out, int ld_h, void* stream);
int unreal_lstm_gates_bwd(int rows, const float* dh_above, const float* dh_rec, float* dc_io, const float* gates_act,
                          const float* c_prev, const float* c_new, float* dpre, float* dpre_absmax0 /*nullable*/,
                          float* dpre_absmax1 /*nullable*/, void* stream);

/* ---- heads, sampling, losses (model/model.py:358-377, 473-516, 559-576; train/trainer.py:147-148) -- */
int unreal_linear_small_fwd(int rows, int K, int NOUT, const float* X, int ldx, const float* W, const float* b,
                            float* out, int ldo, void* stream);
/* dX (nullable) (+)= dO W^T; dW[k*dw_stride_k + n*dw_stride_n] += sum_rows X[row][k] dO[row][n]
 * (strides 0,0 = the natural [K][NOUT] layout; W may be null when dX is null); db (nullable) += sum_rows dO */
int unreal_linear_small_bwd(int rows, int K, int NOUT, const float* X, int ldx, const float* dO, int ldo,
                            const float* W, float* dX, int lddx, int accumulate_dx, float* dW, int dw_stride_k,
                            int dw_stride_n, float* db, void* stream);
int unreal_softmax_sample(int rows, int A, float* logits_pi, int ld, const double* u, int* action, void* stream);
/* one rollout step of the policy in one launch: pi = softmax(X Wp + bp), v = X Wv + bv, action ~ pi (u null: arg max);
 * bit-identical to unreal_linear_small_fwd x2 + unreal_softmax_sample (K = 256 features, A in {3, 4, 6}) */
int unreal_policy_step(int rows, int A, const float* X, int ldx, const float* Wp, const float* bp, const float* Wv,
                       const float* bv, const double* u, float* pi_out, float* v_out, int* action, void* stream);
int unreal_base_loss_grad(int rows, int A, const float* pi, int ld_pi, const float* v, const int* action,
                          const float* adv, const float* R, const int* active, float entropy_beta, float grad_scale,
                          float* dlogits, float* dv, float* losses /*[3]: policy, value, entropy*/, void* stream);
int unreal_vr_loss_grad(int rows, const float* v, const float* R, const int* mask, float grad_scale, float* dv,
                        float* loss, void* stream);
int unreal_rp_loss_grad(int rows, const float* logits, const int* cls, float grad_scale, float* prob, float* dlogits,
                        float* loss, void* stream);
int unreal_colsum(int rows, int cols, const float* X, int ld, float* out, void* stream);
int unreal_relu_mask(int rows, int cols, float* d, int ldd, const float* src, int lds, void* stream);

/* ---- pixel-control head (model/model.py:411-443, 542-557, 805-820) -------------------------------- */
/* hp [N][2592] = relu(pc_fc1); both deconvolutions + dueling combine on the fp16 matrix cores with fp16 hi + lo operands
 * (round 3): hp_absmax = absmax slot covering hp (committed by the pc_fc1 GEMM, c_absmax of unreal_gemm_f32_split_nt).
 * Bootstrap mode (qmax != NULL): max_a Q per cell.  Training mode (d_dec != NULL): loss + dL/d(pre-activation) of both
 * deconvs, and ddec_absmax (nullable slot) receives an upper bound of max |d_dec| -- the scale unreal_pc_deconv_bwd needs. */
int unreal_pc_deconv_fwd(int N, int A, const float* hp, const float* hp_absmax, const float* Wv, const float* bv,
                         const float* Wa, const float* ba, float* qmax, const int* action, const float* target,
                         const int* mask, float lambda, float grad_scale, float* d_dec, float* ddec_absmax, float* loss,
                         void* stream);
int unreal_pc_deconv_bwd(int N, int A, const float* hp, const float* hp_absmax, const float* d_dec, const float* ddec_absmax,
                         const float* Wv, const float* Wa, float* d_hp, float* dhp_absmax /*nullable absmax slot: max |d_hp|*/,
                         float* dWv, float* dbv, float* dWa, float* dba, void* stream);
/* The training pass of the head (model.py:411-443 forward, 542-557 loss, and their gradients) in ONE launch: what
 * unreal_pc_deconv_fwd (training mode) followed by unreal_pc_deconv_bwd computes, with d_dec kept on chip -- its fp16 hi + lo
 * planes take each FRAME's own power-of-two scale (max |dL/dQ| of the frame) instead of the launch's.  *loss and the four
 * parameter gradients are accumulated; d_dec is a nullable [N][400][1+A] output for inspection. */
int unreal_pc_deconv_train(int N, int A, const float* hp, const float* hp_absmax, const float* Wv, const float* bv,
                           const float* Wa, const float* ba, const int* action, const float* target, const int* mask,
                           float lambda, float grad_scale, float* loss, float* d_hp,
                           float* dhp_absmax /*nullable absmax slot: max |d_hp|*/, float* dWv, float* dbv, float* dWa,
                           float* dba, float* d_dec, void* stream);

/* ---- optimiser (train/rmsprop_applier.py:38-43, 83-93, 121) ---------------------------------------- */
int unreal_grad_norm(const float* grad, long n, float* scratch /*256 floats*/, float* norm_out, void* stream);
int unreal_rmsprop_step(float* var, float* ms, float* mom, const float* grad, long n, float lr, float decay,
                        float momentum, float eps, float clip_norm, const float* norm, void* stream);

/* ---- device-to-device hand-over of 4-byte words (start_lstm_state = base_lstm_state_out, trainer.py:228-230; the
 * sampled index lists): an ordinary kernel, so the whole path can run under rocprofv3 --pmc ---------------------- */
int unreal_copy_words(long n, const void* src, void* dst, void* stream);
/* y += alpha * x: the per-call mean of the loss scalars over the G sequential updates of a grouped process() */
int unreal_axpy_f32(long n, float alpha, const float* x, float* y, void* stream);

#ifdef __cplusplus
}
#endif
#endif
