/* libams_hip.so -- C ABI of the MI355X (gfx950) separation hot path.
 *
 * Drop-in boundary for Totoketchup/Adaptive-MultiSpeaker-Separation.  The reference has no FFI: its
 * "operators" are TF-1.x graph ops called from Python.  Each entry point below replaces the
 * TF op call sites named in its comment (paths relative to the reference root); the Python host mirror
 * (adaptive-multispeaker-separation_amd/{models,utils}) binds them through ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 row-major data unless the type says otherwise;
 *   - `stream` is a hipStream_t; work is only enqueued, never synchronised; nothing is allocated;
 *   - the caller owns inputs, outputs and workspaces (sizes from the *_workspace_bytes / *_floats helpers);
 *   - return value: AMS_OK or a negative ams_status; on AMS_E_LAUNCH_FAILED the hipError_t is in ams_last_error();
 *   - re-entrant across streams/threads.  State behind this ABI, all of it: the thread-local last error; ONE process-wide selector of
 *     the product arithmetic (ams_gemm_set_arith, a test / A-B switch; default from AMS_GEMM_X6, read once); a once-per-process cache of
 *     device properties and of the AMS_* tuning environment.  Operand bounds (fp16x3) and the residency cap of a product are ARGUMENTS
 *     of the entry points, not state (ABI 2; ABI 1 had thread-local one-shot setters for both); so is the stream-K scratch (ABI 3).
 *
 * Environment read by the library -- the COMPLETE list (tests/test_abi.py greps csrc/ for getenv and holds this block to it).  Every
 * variable is read ONCE per process (function-local static), never on the launch path, selects between code paths that the tests hold
 * to the same oracle, and exists for A/B measurements and fault-injection tests; a deployment sets none of them.
 *   products (csrc/gemm.hip)   AMS_GEMM_X6 (initial value of ams_gemm_set_arith: 0 = native f32 MFMA), AMS_GEMM_F16X3 (0 = ignore operand
 *                              bounds: bf16x6), AMS_GEMM_SK (stream-K 0/1/2), AMS_GEMM_CVEC (0 = dword epilogue stores), AMS_GEMM_X6CFG,
 *                              AMS_GEMM_X6RULE, AMS_GEMM_X6WASTE (tile choice), AMS_X6_PERSIST (0 = one tile per workgroup),
 *                              AMS_GEMM_SPLITS, AMS_GEMM_GROUP_M (split-K / band height overrides), AMS_GEMM_NOVEC (force the dword-fetch
 *                              f32 kernel), AMS_GEMM_NOPRIO, AMS_MAXPOOL_CFG (0 = 128x128 tile for the fused conv + max-pool),
 *                              AMS_GATHER_LDS (0 = register form of ams_gather_filter_grad), AMS_MAXPOOL_PS (0 = path B's product cuts its operands
 *                              in the kernel also where the pre-split form applies)
 *   recurrence (lstm*.hip)     AMS_LSTM_RING_X6, AMS_LSTM_RING_F16, AMS_LSTM_RING_BWD_F16 (arithmetic of the rings' recurrent products),
 *                              AMS_LSTM_RING_SAFE (write-through hand-off), AMS_LSTM_RING_CUS (pretend a smaller device: fallback tests),
 *                              AMS_LSTM_XCD, AMS_LSTM_FWD_PIPE (per-step fallback kernels: grid order, fetch pipelining)
 *   losses / k-means           AMS_DPCL_LDS (1 = LDS-staged DPCL passes), AMS_DPCL_GRAM_F16 (0 = the fused forward's Gram on the f32 MFMA),
 *                              AMS_KM_TRIES (0 = one workgroup per try), AMS_KM_SOFT (0 = soft accumulation inside kmeans_pass_kernel)
 */
#ifndef AMS_H
#define AMS_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AMS_ABI_VERSION 5

typedef int32_t ams_status;
#define AMS_OK 0
#define AMS_E_INVALID_ARG (-1)
#define AMS_E_WORKSPACE_TOO_SMALL (-2)
#define AMS_E_LAUNCH_FAILED (-3)

int ams_last_error(void);
int ams_abi_version(void);

/* ---- K1  effective filter  f[k,n] = |w[k]| * bases[k,n]      models/adapt.py:104-106, :232-234 ---- */
ams_status ams_front_filter_fwd(const float* w, const float* bases, float* f, int W, int N, void* stream);
ams_status ams_front_filter_bwd(const float* w, const float* bases, const float* df, float* dw, float* dbases, int W, int N,
                                void* stream);

/* ---- K2  analysis filterbank, path A: tf.nn.conv2d stride=hop SAME        models/adapt.py:122 ----
 * x [Bt,L], f [W,N] -> y [Bt,T',N], T' = ceil(L/hop), pad_left = ((T'-1)hop+W-L)/2. */
size_t ams_front_conv_fwd_workspace_bytes(int Bt, int L, int W, int N, int hop);
/* amax_x / amax_f, lds_pad, ws, sk_scratch: see ams_gemm_f32.  amax_y (optional): the launch leaves max |y| there -- the operand
 * bound of the product that reads y, without a pass over y -- only when ams_front_conv_fwd_measures_output(same x, f, geometry) != 0:
 * the 16-bit-pipe form, which needs 16-byte addressable operands (L, hop, pad_left, N, W multiples of 4); the scalar-fetch f32 form
 * such launches fall back to REJECTS amax_y (AMS_E_INVALID_ARG) -- ask first, measure y with ams_absmax_f32 otherwise (ABI 4: the
 * query takes the launch's operands; ABI 3's took none and said yes for shapes the launch then refused). */
ams_status ams_front_conv_fwd(const float* x, const float* f, float* y, int Bt, int L, int W, int N, int hop, const float* amax_x,
                              const float* amax_f, float* amax_y, int lds_pad, void* ws, size_t ws_bytes, void* sk_scratch,
                              size_t sk_bytes, void* stream);
int ams_front_conv_fwd_measures_output(const float* x, const float* f, int L, int W, int N, int hop);
size_t ams_front_conv_bwd_filter_workspace_bytes(int Bt, int L, int W, int N, int hop);
ams_status ams_front_conv_bwd_filter(const float* x, const float* dy, float* df, int Bt, int L, int W, int N, int hop, void* ws,
                                     size_t ws_bytes, void* stream);

/* ---- K3/K4/K5 path B (--with_max_pool): stride-1 conv + tf.nn.max_pool_with_argmax fused (models/adapt.py:115-117);
 * argmax int64 = l*N + n (no batch term).  Sparse (unpool-free) synthesis and gather-form filter
 * gradients (models/adapt.py:210-243, utils/ops.py:94-120). ---- */
size_t ams_front_maxpool_workspace_bytes(int Bt, int L, int N);
size_t ams_front_maxpool_workspace_bytes_w(int Bt, int L, int N, int W);   /* + room for the zero-padded signal copy (faster product) and, where
 * the pre-split form applies (L % 128 == 0, N % 4 == 0), for its operand images: eight shifted fp16 hi | lo copies of the signals = 8 x their
 * bytes (csrc/gemm_ps.hip).  ws must then be 256-byte aligned, else the launch cuts its operands inside the product as before */
/* amax_x / amax_f (optional, both or neither): device pointers to upper bounds of max |x|, max |f| -> the product runs as fp16x3 (with the
 * larger workspace: from images cut once per launch; both forms give the same bits) */
ams_status ams_front_maxpool_fwd(const float* x, const float* f, float* y, long long* argmax, int Bt, int L, int W, int N, int P, int hop,
                                 const float* amax_x, const float* amax_f, void* ws, size_t ws_bytes, void* stream);
/* the sparse kernels take int32 sample positions (argmax / N, converted once) and the synthesis filter TRANSPOSED, f2t [N, W] */
ams_status ams_argmax_to_pos(const long long* argmax, int32_t* pos, long count, int N, void* stream);
ams_status ams_transpose_f32(const float* in, float* out, int rows, int cols, void* stream);
size_t ams_gather_filter_grad_workspace_bytes(int R, int W, int N);
ams_status ams_gather_filter_grad(const float* x, const float* v, const int32_t* pos, float* df, int R, int L, int W, int N, int T,
                                  int rdiv, void* ws, size_t ws_bytes, void* stream);
ams_status ams_synth_unpool_fwd(const float* vals, const int32_t* pos, const float* f2t, float* out, int R, int L, int W, int N, int T,
                                int P, int hop, int S, void* stream);
ams_status ams_synth_unpool_bwd_vals(const float* dout, const int32_t* pos, const float* f2t, float* dvals, int R, int L, int W, int N,
                                     int T, int S, void* stream);

/* ---- dense contractions (tf.matmul / tf.nn.conv1d k=1 / dynamic_rnn input projection) ----
 * C[M,N] (+)= op(A) . op(B) (+ bias[N]);  transX = 0: row-major [rows,cols] as written, 1: stored transposed.
 * mask_period/mask_skip (transA only): reduction rows k with k % period == skip are treated as zero
 * (used for the time-shifted h_{t-1}^T . da product).  utils/ops.py:366-383, :501-503. */
/* Every product entry point takes, besides its operands:
 *   amax_a / amax_b  (optional, both or neither) device pointers to ONE float each, an upper bound of max |value| over the WHOLE A / B
 *                    operand as the entry point sees it (all batches).  Both non-NULL and a launch on the 16-byte-fetch path: "fp16x3"
 *                    arithmetic -- each operand is scaled by 2^(13 - floor(log2(bound))), split exactly into two fp16 terms (22
 *                    significant bits for entries within 2^17 of the bound, an absolute error of bound * 2^-39 below), three fp16 MFMA
 *                    products per f32 product, accumulators unscaled: f32 results at the error level of bf16x6 and of the f32 MFMA
 *                    (tests/test_gpu_gemm_f16.py holds all three against float64; tests/test_gpu_rowwise.py row by row on the operands
 *                    of a real training step), at half the matrix-pipe work.  A bound below the true maximum by more than 4x overflows
 *                    fp16 and yields Inf/NaN (loud); a bound too high by up to 2^10 costs nothing; 0, Inf or NaN select scale 1.
 *                    NULL: bf16x6 (below), which needs no bounds.
 *   lds_pad          0 for a product on the critical path.  > 0: the product is meant to run BESIDE the latency-bound recurrence on
 *                    another stream: that many bytes of unused dynamic LDS per workgroup cap its CU occupancy, and it takes the
 *                    single-accumulator kernel variant (register budget of a CU shared with a ring workgroup); results of the two
 *                    variants differ at the 1e-7 level (tests/test_gpu_gemm_f16.py::test_fp16x3_weight_gradient_bias_capped_and_free).
 *   ws, ws_bytes     split-K partial slabs, ams_gemm_workspace_bytes(M, N, K, nbatch, lds_pad); NULL = no split-K.  A workspace sized
 *                    under another setting is never an error: a launch uses as many slabs as it holds (down to none).
 *   sk_scratch, sk_bytes   (ABI 3; optional) ams_gemm_sk_scratch_bytes() bytes the launch may use for STREAM-K: instead of whole-tile
 *                    rounds plus split-K slabs, every resident workgroup takes an equal share of the k-tiles of the tiles that do not
 *                    fill a round, partial tiles meet in this scratch inside the launch (fixed order: deterministic) -- no slabs, no
 *                    reduce launch, no idle tail round (csrc/gemm.hip: x6_body).  Contract: the first 8192 bytes are ZERO when the
 *                    first launch sees them and every launch leaves them zero; launches sharing one scratch must be ordered on one
 *                    stream (one scratch per stream).  NULL: whole tiles / split-K as before.  AMS_GEMM_SK=0 ignores it.
 * Arithmetic without bounds (process-wide; default 1, or AMS_GEMM_X6 read once; ams_gemm_set_arith for tests and A/B runs): 1 =
 * "bf16x6" -- both f32 operands are split EXACTLY into three bf16 terms (hi + mid + lo, round-to-nearest) and six of the nine bf16 x
 * bf16 partial products (all but mid.lo, lo.mid, lo.lo <= 2^-26 |a.b|) are accumulated in f32 on v_mfma_f32_32x32x16_bf16, which gfx950
 * issues at 16x the rate of its f32 MFMA; f32-level error (tests/test_gpu_gemm_x6.py: against float64, next to mode 0).  0 = native
 * v_mfma_f32_32x32x2_f32 everywhere (bounds are then ignored).  Launches whose operands are not 16-byte addressable use mode 0 whatever
 * the setting.  Inf / NaN operands give NaN in mode 1 (inf - inf in the split) where mode 0 propagates Inf.
 * Replaces nothing in the reference beyond tf.matmul / conv1d / conv2d in f32: it is how those products are issued. */
size_t ams_gemm_workspace_bytes(int M, int N, int K, int nbatch, int lds_pad);
size_t ams_gemm_sk_scratch_bytes(void);
void ams_gemm_set_arith(int mode);
int ams_gemm_get_arith(void);
ams_status ams_gemm_f32(int transA, int transB, int M, int N, int K, const float* A, long lda, const float* B, long ldb, float* C,
                        long ldc, const float* bias, int accumulate, int mask_period, int mask_skip, const float* amax_a,
                        const float* amax_b, int lds_pad, void* ws, size_t ws_bytes, void* sk_scratch, size_t sk_bytes, void* stream);
/* ---- products from PRE-SPLIT fp16 operand images (ABI 4; csrc/gemm_ps.hip) ----
 * The fp16x3 arithmetic of ams_gemm_f32 with the cut made ONCE per operand and step by whoever writes the operand, instead of once per
 * element and workgroup inside the product: C[M,N] = A[M,K] . B[N,K]^T (+ bias[N]) with A and B handed over as "PS32" images.
 *   image of X [R, K] (k contiguous), bound b:  s = 2^(13 - floor(log2 b));  row r = ams_ps_image_pitch(K) bytes (K rounded up to 32
 *   floats' worth: the size of the f32 row); k-tile t of a row = 128 bytes = 32 x fp16 hi | 32 x fp16 lo, hi = fp16(x s),
 *   lo = fp16(x s - hi); k >= K zero.  16-byte aligned.  The bound rule is ams_gemm_f32's (true max below 4 x bound, else Inf/NaN).
 * ams_ps_pack_rows: image of x [R, K] (row pitch ldx floats).  ams_ps_pack_cols: image of w^T for w [K, N] row-major (row pitch ldw):
 * image row n = w[:, n] -- what a forward product x . w needs of its weights.  amax: device pointer to the bound; ams_gemm_ps must be
 * given the SAME values (it undoes both scales on the accumulators).  Results agree with ams_gemm_f32's fp16x3 form to the last bits
 * (same terms, same products, same two accumulator sets; the k order differs).  N, ldc multiples of 4; C, bias 16-byte aligned;
 * images below 2 GB.  Replaces: utils/ops.py:366-383 (dynamic_rnn input projection), :501-503 (conv1d k = 1) in the forward pass. */
size_t ams_ps_image_pitch(int K);
size_t ams_ps_image_bytes(int rows, int K);
ams_status ams_ps_pack_rows(const float* x, long ldx, void* img, int R, int K, const float* amax, void* stream);
ams_status ams_ps_pack_cols(const float* w, long ldw, void* img, int K, int N, const float* amax, void* stream);
ams_status ams_gemm_ps(int M, int N, int K, const void* A_img, const void* B_img, float* C, long ldc, const float* bias,
                       const float* amax_a, const float* amax_b, void* stream);

/* C[M,N] (+)= A^T . B with A stored [K, M] and B [K, N], AND bsum_out[N] (+)= column sums of B in the same pass over B: the
 * weight and bias gradients of Conv1D (utils/ops.py:501-503) and of a BLSTM layer's input kernels from one read of dY / dZ.  M, N, lda,
 * ldb multiples of 4, 16-byte aligned operands; bsum_ws = 32 * N floats of scratch (16-byte aligned). */
ams_status ams_gemm_f32_at_b_colsum(int M, int N, int K, const float* A, long lda, const float* B, long ldb, float* C, long ldc,
                                    int accumulate, float* bsum_out, int bsum_accumulate, float* bsum_ws, const float* amax_a,
                                    const float* amax_b, int lds_pad, void* ws, size_t ws_bytes, void* sk_scratch, size_t sk_bytes,
                                    void* stream);
/* nbatch products of ONE shape in one launch; operand z lives at A + z*a_zs, B + z*b_zs, C + z*c_zs (element offsets, any
 * sign).  No bias.  Used for the two BLSTM directions' recurrent-kernel gradients (h_prev^T . dZ). */
ams_status ams_gemm_f32_batched(int transA, int transB, int M, int N, int K, const float* A, long lda, long a_zs, const float* B,
                                long ldb, long b_zs, float* C, long ldc, long c_zs, int nbatch, int accumulate, int mask_period,
                                int mask_skip, const float* amax_a, const float* amax_b, int lds_pad, void* ws, size_t ws_bytes,
                                void* sk_scratch, size_t sk_bytes, void* stream);
/* Range audit of an fp16x3 operand x [rows, cols] (row pitch ld) against the bound it would be scaled with: out3[0] += the number of
   non-zero entries below bound * 2^-17 (they keep fewer than 22 bits), out3[1] += their energy, out3[2] += the operand's energy
   (the caller zeroes out3).  ops.f16_audit runs it over the operands of a whole step and sends a product class back to bf16x6
   when the estimated relative error of the lost bits exceeds the f32 level (DESIGN 4.0a "guard"). */
ams_status ams_range_share(const float* x, long rows, long cols, long ld, const float* bound, float* out3, void* stream);
/* out[0] = max |x[i]|, i < n, as a float (NaN if any x is NaN): an operand bound for the products above, for operands whose producer
   does not supply one.  Two stream-ordered launches (a 4-byte clear, the reduction); out is a device pointer. */
ams_status ams_absmax_f32(const float* x, long n, float* out, void* stream);

/* ---- K7  dominant-speaker masks: one_hot(argmax_s |rep|, S, a, b)   models/network.py:377-378, :501-502 ----
 * rep_non_mix rows are (b,s) row-major, each TF long; Y [B,TF,S]; argmax [B,TF] int32 (may be NULL). */
ams_status ams_make_masks(const float* rep_non_mix, float* Y, int32_t* argmax, int B, int S, long TF, float a, float b,
                          int take_abs, void* stream);

/* ---- K10  BLSTM recurrence (tf.nn.dynamic_rnn(BasicLSTMCell) x 2 directions)   utils/ops.py:358-383 ----
 * G [B,T,2,4H]: in = x.Wx + b (both directions), out = activated gates (fwd) / d pre-activation (bwd).
 * out [B,T,2H], cst [B,T,2,H].  Uf/Ub = rows D.. of each direction's [D+H,4H] kernel, ldu = 4H.
 * pack: scratch of ams_blstm_pack_floats(H, backward) floats. */
size_t ams_blstm_pack_floats(int H, int backward);
ams_status ams_blstm_recurrent_fwd(float* G, float* out, float* cst, const float* Uf, const float* Ub, long ldu, float* pack,
                                   int B, int T, int H, void* stream);
ams_status ams_blstm_recurrent_bwd(float* G, const float* cst, const float* dout, float* dc, const float* Uf, const float* Ub,
                                   long ldu, float* pack, int B, int T, int H, void* stream);
ams_status ams_blstm_pack(const float* Uf, const float* Ub, long ldu, float* pack, int H, int backward, void* stream);
/* The per-step recurrence with DropoutWrapper's STATE dropout (utils/ops.py:363,373,379, --recurrent_dropout / --recurrent_dropout_enhance
 * != 0, training only; TensorFlow 1.4 masks BOTH parts of the LSTMStateTuple): out / cst keep the cell's own h_t / c_t, hs [B,T,2H] and
 * cs [B,T,2,H] receive the masked states h_t.mh_t / c_t.mc_t the next step starts from; mh, mc [B,T,2,H] are keep-masks scaled by 1/keep,
 * drawn by the caller (TensorFlow's mask stream is not reproducible; the distribution is).  The wrapper's input and output dropout are
 * elementwise on x (one mask per direction) and on the layer output and stay with the caller.  The recurrent-kernel gradients are
 * products of hs (not out) with da.  dc: [B,2,H] workspace as for ams_blstm_recurrent_bwd. */
ams_status ams_blstm_recurrent_fwd_dropout(float* G, float* out, float* cst, float* hs, float* cs, const float* mh, const float* mc,
                                           const float* Uf, const float* Ub, long ldu, float* pack, int B, int T, int H, void* stream);
ams_status ams_blstm_recurrent_bwd_dropout(float* G, const float* cst, const float* cs, const float* dout, float* dc, const float* mh,
                                           const float* mc, const float* Uf, const float* Ub, long ldu, float* pack, int B, int T, int H,
                                           void* stream);
/* Ring form of the same recurrence (csrc/lstm_ring.hip), the default: one launch per layer and pass; a chain = (direction,
 * 16-row batch tile) is a ring of ceil(H/12) resident workgroups on ONE XCD (verified in-launch through HW_REG_XCC_ID; chains
 * whose members do not share an L2, or safe != 0, use write-through stores instead of plain ones).  Forward: h_t travels as
 * 16-byte {3 values, step tag} granules; backward: partial dh tiles (reduce-scatter) + one flag per producer.
 * ams_blstm_ring_sync_bytes returns 0 when the shape cannot use it (H > 336, or more than 512 workgroups): callers then use
 * ams_blstm_recurrent_fwd/bwd; so does a device whose CUs cannot hold the whole grid at once (occupancy query, cached).  sync word 0
 * (uint32) is non-zero after the launch if a bounded in-launch wait timed out; sticky_err (may be NULL) is a uint32 the CALLER owns
 * and clears: it is set whenever word 0 is, so ONE read at the caller's next host sync covers every ring launch since its last
 * check -- also the launches inside a replayed hipGraph, whose per-launch sync buffers nobody looks at again.  The optimizers take
 * the same word as `skip_if_set` and leave parameters and slots untouched when it is non-zero, so a step whose recurrence gave up
 * can be repeated on the per-step kernels instead of being trained on.
 * tch [B,T,2,H]: tanh(c_t), written by the forward ring and read by the backward one (cst keeps c_t).  safe: bit 0 forces the
 * write-through hand-off, bit 1 records a per-phase cycle trace in the sync header (tools/ring_anatomy.py), bit 2 says the CALLER has
 * already zeroed the first ams_blstm_ring_sync_head_bytes() of `sync` in stream order (the whole buffer for both rings: the forward's
 * granule tags and the phase bits of the backward's partial tiles start at 0) -- without it every launch is preceded by its own memset node,
 * ~5 us on the critical path of each of the six ring launches of a training step; the host side (ops.py::_RingArena) clears the
 * buffers of a whole pass with ONE memset on the side stream while the pass starts.
 * dbpart (backward, may be NULL): [B,2,4H] receives sum_t d pre-activation[b,t,dir,:]; the bias gradients are its column sums.
 * Replaces the same dynamic_rnn while_loop (utils/ops.py:358-383). */
size_t ams_blstm_ring_sync_bytes(int B, int H, int backward);
size_t ams_blstm_ring_sync_head_bytes(int B, int H, int backward);
/* amax_u (may be NULL): device pointer to an upper bound of max |U| over both recurrent kernels -> the recurrent product h_{t-1} . U runs
   as fp16x3 instead of bf16x6 -- U scaled by a power of two from *amax_u and split exactly into two fp16 terms, h_{t-1} (|h| < 1)
   scaled by 2^13, three fp16 MFMA products, the two cross terms in their own accumulator.  27 instead of 54 MFMAs per wave and step,
   149 instead of 196 VGPRs.  AMS_LSTM_RING_F16=0 or NULL: bf16x6. */
ams_status ams_blstm_ring_fwd(float* G, float* out, float* cst, float* tch, const float* Uf, const float* Ub, long ldu, const float* amax_u,
                              void* sync, size_t sync_bytes, void* sticky_err, int B, int T, int H, int safe, void* stream);
/* Backward: amax_u (may be NULL) as above -> the recurrent product da_t . U^T runs as fp16x3 too, computed transposed (U is the MFMA's A
   operand, da^T the B operand) so that every batch row of da gets its own power-of-two scale from its own maximum: da has no a-priori
   bound, and a shared scale would let a small row lose bits to a large one.  30 MFMAs of ~17 cycles instead of 60 of 32 per wave and
   step.  AMS_LSTM_RING_BWD_F16=0 or NULL: v_mfma_f32_16x16x4_f32.
   Hand-off of both backward forms (round 4): a partial tile carries its validity in the low mantissa bit of its floats (the phase of the
   step that wrote it; 1 ulp per partial), so a consumer waits on the data itself -- one memory hop per step where the flag protocol of
   rounds 2-3 had three (store acknowledge, flag, tile load). */
ams_status ams_blstm_ring_bwd(float* G, const float* cst, const float* tch, const float* dout, float* dbpart, const float* Uf, const float* Ub,
                              long ldu, const float* amax_u, void* sync, size_t sync_bytes, void* sticky_err, int B, int T, int H, int safe,
                              void* stream);

/* ---- K13  tf.nn.l2_normalize over groups of E       utils/ops.py:323-324 ---- */
ams_status ams_l2norm_fwd(const float* u, float* v, float* inv, long rows, int E, void* stream);
ams_status ams_l2norm_bwd(const float* v, const float* inv, const float* dv, float* du, long rows, int E, void* stream);
/* xn = normalise(normalise(u)) in ONE pass, inv = 1/|u|, inv2 = 1/|normalise(u)| -- the embedding network's Normalize layer
   (models/dpcl.py:32) followed by the k-means' own normalisation of its input (models/Kmeans_2.py:56); the same bits as two
   ams_l2norm_fwd calls.  16-byte addressable rows of E = 40, 32, 20 or 8 floats, else AMS_E_INVALID_ARG (make the two calls). */
ams_status ams_l2norm2_fwd(const float* u, float* xn, float* inv, float* inv2, long rows, int E, void* stream);
/* xn = ams_kmeans_normalize(ams_l2norm_fwd(u)) in one pass and with exactly those bits: the Normalize layer of the embedding network
 * (models/dpcl.py:32) followed by the k-means' own normalisation (models/Kmeans_2.py:40-41) on the inference / enhance paths, where
 * nothing else reads the once-normalised tensor.  E in {40, 32, 20, 8}, 16-byte aligned; AMS_E_INVALID_ARG otherwise. */
ams_status ams_l2norm_kmeans_normalize(const float* u, float* xn, long rows, int E, void* stream);

/* column sums (bias gradients) */
size_t ams_colsum_workspace_bytes(long rows, int cols);
ams_status ams_colsum(const float* x, float* out, long rows, int cols, long ld, int accumulate, void* ws, size_t ws_bytes,
                      void* stream);

/* ---- K14  deep-clustering loss      models/dpcl.py:41-87 ----
 * V [B,TF,E], Y [B,TF,S]; out[0] = cost, out[1..3] = the reference's summaries '1','2','3'.
 * bwd: inv != NULL fuses the l2-normalise backward (writes dU), else writes dV; upstream = optional device scalar
 * d loss / d cost. */
size_t ams_dpcl_workspace_bytes(int B, long TF, int E, int S);
ams_status ams_dpcl_loss_fwd(const float* V, const float* Y, float* out, int B, long TF, int E, int S, void* ws, size_t ws_bytes,
                             void* stream);
ams_status ams_dpcl_loss_bwd(const float* V, const float* Y, const float* inv, const float* upstream, float* dU, int B, long TF,
                             int E, int S, const void* ws, void* stream);

/* Fused training form of K13+K14: U [B,TF,E] is the dense output BEFORE tf.nn.l2_normalize (utils/ops.py:323-324);
 * one pass computes inv[B,TF] = 1/max(|u|,1e-6), the loss terms and (V_out != NULL) the normalised embeddings.
 * The backward recomputes v = u*inv and applies d loss/dV and the l2-normalise Jacobian in one pass. */
/* Byte offset, inside the ams_dpcl_loss_fwd_u workspace, of ONE float: max |dU| of the latest ams_dpcl_loss_bwd_u on that workspace
   (cleared by ams_dpcl_loss_fwd_u, raised by every backward since: always an upper bound of the latest dU).  It is the operand
   operand bound (amax_a / amax_b of the product entry points) of the dense layer's dX and dW products, produced without another pass over the 210 MB of dU.
   (ams_dpcl_loss_bwd_u takes the workspace as const: this slot is the one thing it writes there.) */
size_t ams_dpcl_u_amax_offset(int B, long TF, int E, int S);
size_t ams_dpcl_u_workspace_bytes(int B, long TF, int E, int S);
/* The label counts Y^T 1 that the fused forward needs first (fixed-order partial sums, kept in ws): Y is known long before U, so a
   training step may issue them early -- beside the recurrence, on another stream -- and pass counts_ready = 1 to the forward on the
   SAME workspace; with counts_ready = 0 the forward counts by itself. */
ams_status ams_dpcl_u_count_labels(const float* Y, int B, long TF, int E, int S, void* ws, size_t ws_bytes, void* stream);
/* ams_make_masks (K12, without its argmax output) and ams_dpcl_u_count_labels in one pass: a plugged training step builds its labels
   from the per-source representations and counts them as it writes them (same Y, same count bits as the two calls). */
ams_status ams_dpcl_u_make_masks(const float* rep_non_mix, float* Y, int B, int S, long TF, int E, float a, float b, int take_abs,
                                 void* ws, size_t ws_bytes, void* stream);
ams_status ams_dpcl_loss_fwd_u(const float* U, const float* Y, float* inv, float* V_out, float* out, int B, long TF, int E, int S,
                               int counts_ready, void* ws, size_t ws_bytes, void* stream);
ams_status ams_dpcl_loss_bwd_u(const float* U, const float* Y, const float* inv, const float* upstream, float* dU, int B, long TF,
                               int E, int S, const void* ws, void* stream);

/* ---- enhance-layer output stage   models/network.py:640-660 ----
 * u [B,S,TF] = Conv1D output of the enhance stack (rows (b,s)); X [B,TF] mixture representation.
 * cost_in[b,p,s] = act_s(u[b,:,p]) * X[b,p] (act over the SPEAKER axis: 0 none, 1 softmax, 2 tanh); separated (may be NULL)
 * is the same values as [B,S,TF] (network.py:657-660).  bwd: either upstream gradient may be NULL. */
ams_status ams_enhance_output_fwd(const float* u, const float* X, float* cost_in, float* separated, int B, int S, long TF,
                                  int nonlin, void* stream);
ams_status ams_enhance_output_bwd(const float* u, const float* X, const float* d_cost_in, const float* d_separated, float* du, int B,
                                  int S, long TF, int nonlin, void* stream);

/* ---- L41 speaker vectors   models/L41.py:60-68: tf.nn.l2_normalize(speaker_centroids, 1) then gather_nd by I ----
 * table [nspk,E], I int32 [R = B*S] -> vs [R,E]; bwd writes the whole d_table (deterministic order). */
ams_status ams_l41_speaker_fwd(const float* table, const int* I, float* vs, int R, int E, int nspk, int normalize, void* stream);
ams_status ams_l41_speaker_bwd(const float* table, const int* I, const float* d_vs, float* d_table, int R, int E, int nspk,
                               int normalize, void* stream);

/* ---- optional input conditioning / weightings   models/network.py:381-396,409-454,504-521, models/Kmeans_2.py:76-80 ----
 * row = one utterance (T*F values).  ams_row_transform: pre 0 none | 1 abs | 2 sqrt | 3 log10(x+1e-12), then norm 0 none |
 * 1 (z-min)/(max-min) | 2 (z-mean)/sqrt(population var) | 3 z*[max-z < thr] (silence_mask_db/20).
 * ams_weight_masks: y[b,p,:] *= f(|X|/max|X|) (1 linear, 2 sqrt, 3 square) and/or [log10(max/|X|) < sil_thr], in place.
 * ams_silence_weights: w = [log10(max(lat)/lat) < thr]  (k-means 'notsilent' weights). */
ams_status ams_row_transform(const float* x, float* out, int rows, long n, int pre, int norm, float thr, void* stream);
ams_status ams_weight_masks(const float* X, float* y, int B, long TF, int S, int mode, int use_silence, float sil_thr, void* stream);
ams_status ams_silence_weights(const float* lat, float* w, int rows, long n, float thr, void* stream);

/* ---- pre-training oracle separator   models/adapt.py:173-196 ----
 * y [B(1+S), TN] (B mixture rows, then (b,s) source rows) -> out [B*S, TN]; mode 0 'mask' = mix*(nm/mix), 1 'perfect' =
 * mix - (sum of the other sources).  bwd writes dy for all rows. */
ams_status ams_pretrain_separator_fwd(const float* y, float* out, int B, int S, long TN, int mode, void* stream);
ams_status ams_pretrain_separator_bwd(const float* dout, float* dy, int B, int S, long TN, int mode, void* stream);

/* ---- default-on terms of the pre-training objective   models/adapt.py:127-132 (p_hat, sparse_constraint), 310-316 and 377-384
 * (regularization, non-negativity), utils/ops.py:46-54 (kl_div / logfunc); CLI defaults utils/trainer.py:151-161 ----
 * ams_abs_colsum_fwd:       p_hat[M] = sum_b |y[b, m]| over the Bt rows (tf.reduce_sum(tf.abs(y), 0)); fixed slab order.
 * ams_kl_sparsity_fwd:      out[0] = sum_m p log(clip(p)/clip(p_hat)) + (1-p) log(clip(1-p)/clip(1-p_hat)), clip to [1e-10, 1].
 * ams_kl_sparsity_bwd:      dy[Bt, M] (+)= upstream[0] * gscale * sign(y) * d kl / d p_hat (clip_by_value passes the gradient
 *                           inside [1e-10, 1] only); gscale carries the data-parallel world size (p_hat is a batch SUM).
 * ams_negative_energy_fwd:  out[0] = mean_b sum_{t,n} min(y, 0)^2.
 * ams_sumsq_bwd:            dx (+)= upstream[0] * scale * 2 x (mode 0: of ams_sumsq) / * 2 min(x, 0) (mode 1: of the above, with
 *                           scale = 1 / Bt).   upstream is a DEVICE scalar: nothing here synchronises. */
size_t ams_abs_colsum_workspace_bytes(int Bt, long M);
ams_status ams_abs_colsum_fwd(const float* y, float* p_hat, int Bt, long M, void* ws, size_t ws_bytes, void* stream);
ams_status ams_kl_sparsity_fwd(const float* p_hat, float* out, long M, float p, void* ws, size_t ws_bytes, void* stream);
ams_status ams_kl_sparsity_bwd(const float* y, const float* p_hat, const float* upstream, float gscale, float* dy, int Bt, long M,
                               float p, int accumulate, void* stream);
ams_status ams_negative_energy_fwd(const float* y, float* out, int Bt, long M, void* ws, size_t ws_bytes, void* stream);
ams_status ams_sumsq_bwd(const float* x, const float* upstream, float scale, float* dx, long n, int mode, int accumulate, void* stream);

/* ---- K24  optimizers     models/network.py:181-194, utils/ops.py:686-703 ---- */
ams_status ams_opt_amsgrad(float* p, const float* g, float* m, float* v, float* vhat, long n, float lr_t, float beta1,
                           float beta2, float eps, float grad_scale, const void* skip_if_set, void* amax_slots, float* bound_out, const float* grad_scale_dev, void* stream);
ams_status ams_opt_rmsprop(float* p, const float* g, float* ms, long n, float lr, float decay, float eps, float grad_scale,
                           const void* skip_if_set, void* amax_slots, float* bound_out, const float* grad_scale_dev, void* stream);
ams_status ams_opt_momentum(float* p, const float* g, float* accum, long n, float lr, float momentum, float grad_scale,
                            const void* skip_if_set, void* amax_slots, float* bound_out, const float* grad_scale_dev, void* stream);
/* grad_scale_dev (optional): one more factor on the gradients, read from the device -- the tf.clip_by_global_norm factor
 * (models/network.py:185-190) from ams_clip_scale: out[0] = clip / max(sqrt(sumsq[0]) * pre_scale, clip). */
ams_status ams_clip_scale(const float* sumsq, float pre_scale, float clip, float* out, void* stream);
/* amax_slots + bound_out (optional, both or neither; amax_slots = 192 uint32 of scratch, zero before the first use, left zero): the
 * kernels also leave bound_out[0] = max |p| of what they wrote (folded inside the launch by the last block to finish; a step whose update
 * was skipped leaves the bound as it was).  The fp16x3 products of the NEXT step scale the weights by that bound: no pass over the 47 MB
 * of parameters in front of the first product (ops.param_amax; the measuring form is ams_absmax_f32). */
/* One batch into the static input buffers of a captured step: dst[0..n) = src[0..n) (both 16-byte aligned), dst2 = src2 (n2_bytes, may be
 * 0: the speaker indices), and amax_out[0] = max |src| (optional; scratch = ams_stage_inputs_scratch_bytes() bytes, first word zero on
 * first use, left zero).  Replaces three copy launches and the pass that bounded the waveforms for the front product
 * (models/network.py: the feed of inputs/mix_input, non_mix_input, indicies -- reference models/network.py:44-85). */
/* zero two device regions in ONE launch (16-byte aligned, sizes multiples of 16; either may be empty): the gradient buffer and the
 * recurrence rings' sync arena at the head of a training step. */
ams_status ams_zero2(void* a, size_t a_bytes, void* b, size_t b_bytes, void* stream);
size_t ams_stage_inputs_scratch_bytes(void);
ams_status ams_stage_inputs(const float* src, float* dst, long n, const void* src2, void* dst2, long n2_bytes, float* amax_out,
                            void* scratch, void* stream);
ams_status ams_sumsq(const float* x, float* out, long n, void* ws, size_t ws_bytes, void* stream);
/* measurement aid: buf[slot] (uint64) = the device's constant-rate wall clock when the stream reaches this point; ams_stamp_rate() =
 * its ticks per second.  Stamps bracket launches INSIDE a replayed hipGraph (HIP events recorded during capture cannot be read back). */
ams_status ams_stamp(void* buf, int slot, void* stream);
long ams_stamp_rate(void);

/* ---- framed products: out[(r,t), n] = sum_k xpad[r, t*hop + k - pad_left] * Bm[k, n]   (K6 STFT as a DFT product,
 * tf.contrib.signal.stft models/network.py:482-492; also the generic form of K2) ---- */
ams_status ams_frames_matmul(const float* x, const float* Bm, float* out, int R, int L, int W, int N, int hop, int T, int pad_left,
                             void* stream);

size_t ams_frames_matmul_bwd_filter_workspace_bytes(int R, int W, int N, int T);
ams_status ams_frames_matmul_bwd_filter(const float* x, const float* dy, float* dB, int R, int L, int W, int N, int hop, int T,
                                        int pad_left, void* ws, size_t ws_bytes, void* stream);

/* ---- K5/K21 overlap-and-add: out[r,l] = sum_t frames[r,t,l+pad_left-t*hop]
 * second half of tf.nn.conv2d_transpose (models/adapt.py:241-243) and of inverse_stft (models/network.py:598-602) ---- */
ams_status ams_overlap_add(const float* frames, float* out, int R, int T, int W, int L, int hop, int pad_left, void* stream);

/* ---- K22/K23 waveform statistics for SDR / L2 / PIT costs   models/adapt.py:321-372,404-431; network.py:196-221,662-724
 * stats per utterance: D[S*S] (<t_s,a_s'>) | Na[S] | Nt[S] | Tm[S] (<t_s,mix>) | Nm[1];  mix may be NULL ---- */
size_t ams_pair_stats_workspace_bytes(int B, int S, long L);
ams_status ams_pair_stats_fwd(const float* target, const float* est, const float* mix, float* stats, int B, int S, long L, void* ws,
                              size_t ws_bytes, void* stream);
ams_status ams_pair_stats_bwd(const float* target, const float* est, const float* gstats, float* dest, int B, int S, long L,
                              void* stream);
/* costs from the stats table in ONE launch (the [B,S,S] arithmetic of adapt.py:321-372, network.py:662-724), table layout
 * D[S*S] | Q[S*S] (|t_s - a_s'|^2) | Na | Nt | Tm | Nm:  mode 0 pre-training -> out = (mean_b sum_s Q_ss, mean_bs Nt Na/(D_ss^2+1e-12));
 * mode 1 PIT squared error -> out[0] = mean_b min_p red_s(Q[b,s,p(s)] * cl), red = sum (cs = 1) or mean (cs = 1/S);
 * mode 2 Adapt.cost non-pretraining branch -> out = (mean_b min_p sum_s Q/L, mean_i sum_s min_j Nt[i,s] Na[j,s]/(D2[s,i,j]^2+1e-12)),
 * D2 [S,B,B] the cross-batch dot products (adapt.py:361-365).  perms [P,S] int32, lexicographic; pbest [B], jbest [B,S] int32
 * carry the arg-minima from the forward to the backward call; gstats [B,NS] / gD2 [S,B,B] are fully overwritten. */
ams_status ams_pair_combine_fwd(const float* stats, const float* D2, const int* perms, float* out, int* pbest, int* jbest, int B, int S,
                                int P, int mode, float cl, float cs, void* stream);
ams_status ams_pair_combine_bwd(const float* stats, const float* D2, const int* perms, const float* gout, const int* pbest,
                                const int* jbest, float* gstats, float* gD2, int B, int S, int P, int mode, float cl, float cs,
                                void* stream);

/* ---- K20 mask application   models/network.py:577-581 ---- */
ams_status ams_apply_masks_fwd(const float* X, const float* masks, float* sep, int B, int S, long TF, void* stream);
ams_status ams_apply_masks_bwd(const float* X, const float* dsep, float* dmasks, int B, int S, long TF, void* stream);

/* ---- overlap metric of the pretraining separator   models/adapt.py:141-160 ---- */
ams_status ams_overlap_metric_fwd(const float* y, float* out, int B, int S, long TN, void* ws, size_t ws_bytes, void* stream);
ams_status ams_overlap_metric_bwd(const float* y, const float* upstream, float* dy, int B, int S, long TN, void* stream);

/* ---- K6/K7/K21 complex glue around the DFT products   models/network.py:497-499, 589-596 ---- */
/* ri [rows, ld_ri] = [Re(F) | Im(F) | padding]: ld_ri >= 2F floats between rows (the DFT product pads its output rows to a multiple of 4
   floats so that it stays on the 16-byte fetch path: F = W/2 + 1 is odd) */
ams_status ams_cplx_mag_phase(const float* ri, float* mag, float* phasor, long rows, int F, long ld_ri, void* stream);
ams_status ams_cplx_apply_fwd(const float* sep, const float* phasor, float* z, long rows, int F, int S, int T, void* stream);
ams_status ams_cplx_apply_bwd(const float* dz, const float* phasor, float* dsep, long rows, int F, int S, int T, void* stream);

/* ---- K15 L41 loss   models/L41.py:150-178 (sampling=None) ----
 * emb_is_u (all four entry points): 0 = emb holds the embeddings the loss is defined on; 1 = emb is the network output BEFORE
 * tf.nn.l2_normalize over E (models/L41.py:43 Normalize(3), utils/ops.py:323): every point is normalised in registers inside the pass
 * and the backward returns the gradient w.r.t. that un-normalised tensor (the normalise Jacobian applied before the store) -- K13
 * fused into K15 as it is into K14 (ams_dpcl_loss_fwd_u): no l2-normalise pass before the loss and none after it. */
size_t ams_l41_workspace_bytes(int B, long TF, int E, int S);
ams_status ams_l41_loss_fwd(const float* emb, const float* y, const float* vspk, float* cost, int B, long TF, int E, int S, int emb_is_u,
                            void* ws, size_t ws_bytes, void* stream);
/* amax_out (ABI 4; optional, both backward entry points): ONE float that receives max |demb| of the launch -- the operand bound of the
 * two dense-layer products that read demb, without a pass over it (839 MB at cfg5); folded from per-block maxima by the launch's own
 * finishing kernel: nothing to clear, deterministic. */
ams_status ams_l41_loss_bwd(const float* emb, const float* y, const float* vspk, const float* upstream, float* demb, float* dvspk,
                            float* amax_out, int B, long TF, int E, int S, int emb_is_u, void* ws, size_t ws_bytes, void* stream);
/* ... with negative sampling (--sampling K)   models/L41.py:69-147,165-166:
 *   cost[b,t,f] += ns_rate * mean_k -log(sigmoid(-<negs[b, sel, k, :], emb[b,t,f,:]>))
 * negs [B,NSEL,K,E] = rows of the (normalised) speaker table the caller gathered: NSEL = 1 -- one set per utterance, ns_method
 * 'random' (:117-139) -- or NSEL = S -- sel = argmax_s y[b,t,f,s], the K nearest neighbours of the bin's dominant speaker,
 * ns_method 'k-nearest' (:91-116).  K <= 16, NSEL*K <= 32.  dnegs [B,NSEL,K,E] is overwritten. */
size_t ams_l41_ns_workspace_bytes(int B, long TF, int E, int S, int NSEL, int K);
ams_status ams_l41_loss_ns_fwd(const float* emb, const float* y, const float* vspk, const float* negs, float* cost, int B, long TF, int E,
                               int S, int NSEL, int K, float ns_rate, int emb_is_u, void* ws, size_t ws_bytes, void* stream);
ams_status ams_l41_loss_ns_bwd(const float* emb, const float* y, const float* vspk, const float* negs, const float* upstream, float* demb,
                               float* dvspk, float* dnegs, float* amax_out, int B, long TF, int E, int S, int NSEL, int K, float ns_rate,
                               int emb_is_u, void* ws, size_t ws_bytes, void* stream);

/* ---- K16-K19 batched k-means   models/Kmeans_2.py:40-188 ----
 * xn [b,L,E] normalised input (ams_kmeans_normalize); rows r = b_idx*tries + try; centroids [b*tries, C, E];
 * w [b,L] silence weights or NULL; beta < 0 => hard assignment; w_mod_b reproduces the reference's tile order. */
ams_status ams_kmeans_normalize(const float* x, float* xn, long nrows, int E, void* stream);
size_t ams_kmeans_workspace_bytes(int R, long L, int E, int C);
ams_status ams_kmeans_init(const float* xn, const int32_t* init_idx, float* centroids, int b, int tries, long L, int E, int C,
                           void* stream);
ams_status ams_kmeans_iterate(const float* xn, const float* w, const float* cent_in, float* cent_out, float* den_out, int b, int tries,
                              long L, int E, int C, float beta, int w_mod_b, void* ws, size_t ws_bytes, void* tickets, void* stream);
/* tickets (ams_kmeans_iterate / ams_kmeans_assign; optional): b * tries uint32, zero before the first use and left zero -- the chunk
 * partials of a row are then added up INSIDE the pass by the workgroup that stores the row's last one (same chunk order: same bits),
 * instead of by a reduce launch behind each of the nb_steps + 1 passes.
 * Hard passes with E = 40, C = 2 and tries a multiple of 5 (with or without silence weights) serve FIVE tries of an utterance from one read of its points
 * (csrc/kmeans.hip, kmeans_hard_tries_kernel / _final_kernel; AMS_KM_TRIES=0 in the environment: one workgroup per try as elsewhere);
 * every path produces the same bits: the summation order is part of the contract (oracle/kmeans.py: 8192-point chunks, lane j of 256
 * adds its 32 points in sequence, one halving tree per wavefront, wavefront totals in (chunk, wavefront) order).  ams_kmeans_assign with
 * inertia = NULL writes labels only. */
/* Backward of the unrolled soft k-means for b already-selected rows (models/Kmeans_2.py:145-188 under tf.gradients;
 * csrc/kmeans_soft.hip): one call enqueues the final-assignment pass, the n_it iteration passes in reverse (each one streaming read of
 * xn, partial sums finished by the last-arriving workgroup: no reduce launches) and ONE pass that writes dx.
 *   xn [b,L,E] normalised embeddings; w [b,L] silence weights of the iterations or NULL; w_final: weights of the returned assignment
 *   or NULL (--end_assign: all ones); cents [n_it+1,b,C,E] = c_0..c_n; dens [n_it,b,C] = sum_l lab_i; dsel [b,C,E] = d/d c_n or NULL;
 *   dout [b,L,C] = d/d returned soft labels or NULL.  Out: dx [b,L,E] (fully written), g0 [b,C,E] = d/d c_0.
 *   seed [b,C] (may be NULL: the caller scatters g0): the points c_0 was picked from -- g0 is added onto those rows of dx;
 *   inv [b,L] (may be NULL; needs seed): 1/|u| of the tf.nn.l2_normalize that produced xn (Kmeans_2.py:56, utils/ops.py:323) -- dx then
 *   leaves as the gradient w.r.t. u, the Jacobian applied in the pass that holds the point in registers instead of in a pass of its
 *   own; inv0 [b,L] (may be NULL; needs inv): that u was itself normalise(u0) with inv0 = 1/|u0| (ams_l2norm2_fwd: the embedding
 *   network's Normalize layer, models/dpcl.py:32, followed by the k-means' own) -- dx leaves as the gradient w.r.t. u0.
 *   ws: ams_kmeans_soft_bwd_workspace_bytes.  ABI 5: the call also leaves max |dx| (a float, at byte ams_kmeans_soft_bwd_amax_offset of
 *   ws): the operand bound of the dense layer's gradient products, folded by the pass that writes dx instead of a pass over dx. */
size_t ams_kmeans_soft_bwd_workspace_bytes(int b, long L, int E, int C, int n_it);
size_t ams_kmeans_soft_bwd_amax_offset(int b, long L, int E, int C, int n_it);
ams_status ams_kmeans_soft_bwd(const float* xn, const float* w, const float* w_final, const float* cents, const float* dens, const float* dsel,
                               const float* dout, const float* inv, const float* inv0, const int32_t* seed, float* dx, float* g0, int b, long L,
                               int E, int C, float beta, int n_it, void* ws, size_t ws_bytes, void* stream);
ams_status ams_kmeans_assign(const float* xn, const float* w, const float* cent, int32_t* labels, float* soft, float* inertia, int b,
                             int tries, long L, int E, int C, float beta, int w_mod_b, void* ws, size_t ws_bytes, void* tickets, void* stream);
ams_status ams_kmeans_select(const float* inertia, const float* centroids, int32_t* best, float* selected, int b, int tries, int E,
                             int C, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AMS_H */
