/*
 * e3d_hip.h -- C-ABI of the MI355X (gfx950) denoising hot path.
 *
 * The reference (LabJunBMI/E3-invaraint-diffusion-model) is pure Python and exposes no
 * FFI/plugin interface of its own; its seam is the Python call
 * surface of structure_model/{model,sample}.py and sequence_model/{model,sample}.py.  This
 * header is therefore the boundary the reference-side binding (ctypes)
 * would load: plain pointers and sizes, no torch types.  Each entry point cites the
 * reference lines (relative to /root/reference) whose arithmetic it replaces.
 *
 * Conventions
 *   - every pointer is DEVICE memory (HBM), fp32 row-major unless stated, 16-byte aligned;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing
 *     synchronises with the host and nothing is allocated;
 *   - return value: 0 = enqueued, <0 = argument error (message via e3d_last_error()),
 *     >0 = hipError_t from the launch.
 */
#ifndef E3D_HIP_H
#define E3D_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define E3D_ABI_VERSION 4

/* ``terms`` of the split-operand entry points: how an fp32 operand enters the 16-bit matrix cores.
 *   3  (bf16x3): 2 bf16 terms, 3 cross products, ~2^-17 per product, fp32 exponent range;
 *   6  (bf16x6): 3 bf16 terms, 6 cross products, ~2^-24 (fp32 grade), fp32 exponent range;
 *   19 (f16x3):  2 fp16 terms (11 bits each), 3 cross products, ~2^-21 per product -- fp32 grade at the cost of
 *                bf16x3 -- for operands inside the fp16 range: |x| < 65504 (larger values become inf, i.e. the result is
 *                NaN/inf, never silently wrong), elements below 2^-14 carry an ABSOLUTE error of 2^-25.  Forward
 *                (K-contiguous) GEMM layout and the cooperative attention kernel; other paths run bf16x6. */
#define E3D_TERMS_BF16 1   /* plain bf16 products (RNE operands, ~2^-8), fp32 accumulate: the reference's TRAINING precision
                            * (torch.set_float32_matmul_precision("medium"), structure_model/train_model.py:120); GEMM entry
                            * points only, opt-in (E3D_TRAIN_ARITHMETIC=bf16) -- never an inference default */
#define E3D_TERMS_BF16X3 3
#define E3D_TERMS_BF16X6 6
#define E3D_TERMS_F16X3 19

#define E3D_ACT_NONE 0
#define E3D_ACT_GELU 1 /* exact erf GELU: transformers get_activation("gelu"), nn.GELU() */
#define E3D_ACT_SILU 2 /* nn.SiLU in SELayer.adaLN_modulation, structure_model/model.py:34 */

int e3d_abi_version(void);
const char* e3d_last_error(void);

/* out[M,N] = act(A[M,K] @ W[N,K]^T + bias[N]) -- every nn.Linear of the path
 * (BertSelfAttention query/key/value, BertSelfOutput.dense, BertIntermediate, BertOutput:
 * transformers 4.38.2 modeling_bert.py; SELayer.adaLN_modulation / mlp:
 * structure_model/model.py:32-47; predictor dense1: structure_model/model.py:149-151).
 * lda/ldc are row strides in elements.  Requires N % 128 == 0, K % 32 == 0.  bias may be NULL. */
int e3d_gemm_bias_act_f32(const float* A, int64_t lda, const float* W, const float* bias,
                          float* out, int64_t ldc, int M, int N, int K, int act, void* stream);

/* Same contract as e3d_gemm_bias_act_f32, computed on the bf16 matrix cores with each fp32
 * operand split into 2 (terms = 3 cross products) or 3 (terms = 6) bf16 terms and fp32
 * accumulation: fp32-grade results (terms = 6: ~2^-24 per product; terms = 3: ~2^-17) at 6/16 or
 * 3/16 of the exact kernel's MFMA cost. */
int e3d_gemm_bias_act_f32_split(const float* A, int64_t lda, const float* W, const float* bias,
                                float* out, int64_t ldc, int M, int N, int K, int act, int terms,
                                void* stream);

/* e3d_gemm_bias_act_f32_split that also reports the largest |out| it wrote: ``out_absmax`` (device, one float, may be
 * NULL) is raised atomically to max(*out_absmax, max |out[m][n]|) -- never lowered, so the caller zeroes it when it wants
 * a fresh figure; a NaN / inf output leaves a NaN / inf there.  act must be E3D_ACT_NONE when out_absmax is given.
 * Purpose: the bounds of e3d_relkey_attn_fwd_split_ex (below), which let the attention kernels skip all-padding key
 * tiles only when that is provably exact.  Costs two integer VALU operations per output element.
 * ``out_scale``: out = act(out_scale * (A W^T) + bias).  For terms = 19 (f16x3) a caller hands in W * 2^k (exact) with k
 * chosen so that max |W| 2^k sits near 2^12, and out_scale = 2^-k (exact): both fp16 terms of every weight element down
 * to 2^-15 of the largest are then normal numbers (22 bits), instead of an absolute 2^-25 floor below 2^-14 -- a
 * uniformly tiny weight (|w| ~ 1e-6) goes from ~2e-2 to fp32-grade relative error -- and a weight beyond the fp16 range
 * cannot overflow.  1.0f = the plain product (bit-identical to e3d_gemm_bias_act_f32_split). */
int e3d_gemm_bias_act_f32_split_ex(const float* A, int64_t lda, const float* W, const float* bias, float* out,
                                   int64_t ldc, int M, int N, int K, int act, int terms, float* out_absmax,
                                   float out_scale, void* stream);
/* target = max(target, max_i |x[i]|), same conventions (distance-embedding tables, test aids). */
int e3d_absmax_f32(const float* x, int64_t n, float* target, void* stream);

/* Fused attention, head dim 64 ("edge aggregation" of the north star):
 *   S = (Q K^T + R) / sqrt(64) + (1 - key_mask) * -10000;  out = softmax(S) V
 *   R[l,r] = q_l . dist_emb[l - r + P - 1]   (relative_key, only when dist_emb != NULL)
 * Replaces transformers 4.38.2 BertSelfAttention.forward as called from
 * structure_model/model.py:40,62,197-213 and sequence_model/model.py:39,226-235, plus
 * _exetend_attention_mask (structure_model/model.py:226-231).
 *   q: token (b,l) head h at q + b*q_bs + l*q_rs + h*64 (element strides), same for k, v;
 *   key_mask [B,Lk] holds 1.0/0.0 (NULL = all ones); out [B,Lq,nh*64];
 *   lse (optional, [B,nh,Lq]) receives log-sum-exp of S rows for the backward pass.
 * Requires Lq,Lk >= 1; with dist_emb: Lq == Lk <= P and dist_emb is [2P-1,64]. */
int e3d_relkey_attn_fwd(const float* q, int64_t q_bs, int64_t q_rs,
                        const float* k, int64_t k_bs, int64_t k_rs,
                        const float* v, int64_t v_bs, int64_t v_rs,
                        const float* dist_emb, int P, const float* key_mask,
                        float* out, float* lse, int B, int nh, int Lq, int Lk, void* stream);

/* Same contract as e3d_relkey_attn_fwd on the bf16 matrix cores: K, E, Q, V and the softmax
 * probabilities enter the MFMAs as 2 (terms = 3) or 3 (terms = 6, fp32-grade) bf16 split terms,
 * accumulation, softmax and the rel-key skew stay fp32.
 * Softmax exponential: exp(x) is evaluated as exp2(x * log2(e)) on the hardware v_exp_f32 (<= 1 ulp of fp32
 * exp2; the product x * log2(e) rounds once more), not libm expf -- inside the 1e-4 contract
 * (tests/test_kernels_gpu.py), stated here because SURVEY H2 asks for it.  The exact-fp32 entry point above
 * (e3d_relkey_attn_fwd) uses libm expf.
 * This entry point takes no scratch and never allocates: rel-key calls run the per-wave kernel.  The faster
 * workgroup-cooperative kernel needs the distance table as bf16 planes in caller-provided memory -- bind
 * e3d_relkey_attn_fwd_split_ex (below) for that. */
int e3d_relkey_attn_fwd_split(const float* q, int64_t q_bs, int64_t q_rs,
                              const float* k, int64_t k_bs, int64_t k_rs,
                              const float* v, int64_t v_bs, int64_t v_rs,
                              const float* dist_emb, int P, const float* key_mask,
                              float* out, float* lse, int B, int nh, int Lq, int Lk, int terms,
                              void* stream);

/* The same product for small launches (one to ~16 pockets per step: BASELINE configs[0]; M <= 4096 accepted, the host mirror
 * sends up to ~768 output tiles of 32x32 here, where it beats the tiled kernels), cut along K as well so
 * that the whole chip streams the weight: one wave per (32 rows, 32 columns, K slice) writes a partial tile into the
 * workspace, a second small launch sums the slices in slice order (deterministic) and applies bias and activation.
 * Two launches on the given stream.  terms = 3 (bf16x3) or 19 (f16x3).  N % 32 == 0, K % 16 == 0, lda % 4 == 0,
 * ldc % 4 == 0; A, W, bias, out 16-byte aligned.
 * ``workspace``: device memory of >= e3d_gemm_skinny_workspace_bytes(M, N, K) bytes, 16-byte aligned, no initialisation
 * needed, not shared by launches that may run concurrently (one per stream). */
int64_t e3d_gemm_skinny_workspace_bytes(int M, int N, int K);
int e3d_gemm_skinny_f32_split(const float* A, int64_t lda, const float* W, const float* bias, float* out,
                              int64_t ldc, int M, int N, int K, int act, int terms, void* workspace,
                              int64_t workspace_bytes, void* stream);
/* ... with the |out| maximum and the accumulator scale of e3d_gemm_bias_act_f32_split_ex (``out_absmax`` may be NULL;
 * act = none when given; out_scale = 1.0f: the plain product). */
int e3d_gemm_skinny_f32_split_ex(const float* A, int64_t lda, const float* W, const float* bias, float* out,
                                 int64_t ldc, int M, int N, int K, int act, int terms, void* workspace,
                                 int64_t workspace_bytes, float* out_absmax, float out_scale, void* stream);
/* Diagnostic switch (A/B timing, tools/lab/skinny_ab.py): the K-slicing plan of the kernels above -- waves per CU the
 * plan aims for (times 2; default 3 = 1.5 per CU) and the shortest K slice (default 96).  Query the workspace size AFTER
 * changing the plan.  Results differ between plans only by fp32 summation order. */
void e3d_gemm_skinny_plan_select(int waves_per_cu_x2, int min_k_slice);
/* BertSelfOutput / BertOutput in one call (transformers 4.38.2 modeling_bert.py: dense -> dropout(eval: identity) ->
 * LayerNorm(hidden + input)):  out[M,H] = LayerNorm(A . W^T + bias + residual) * gamma + beta, the second launch of the
 * product above doing the row finish.  Bit-identical to e3d_gemm_skinny_f32_split followed by
 * e3d_residual_layernorm_fwd.  H in {256, 512, 768, 1024}; out is contiguous [M,H]; residual may be NULL. */
int e3d_gemm_skinny_residual_layernorm_f32_split(const float* A, int64_t lda, const float* W, const float* bias,
                                                 const float* residual, const float* gamma, const float* beta, float eps,
                                                 float* out, int M, int H, int K, int terms, void* workspace,
                                                 int64_t workspace_bytes, void* stream);
/* ... with out_scale: LayerNorm(out_scale * (A W^T) + bias + residual). */
int e3d_gemm_skinny_residual_layernorm_f32_split_ex(const float* A, int64_t lda, const float* W, const float* bias,
                                                    const float* residual, const float* gamma, const float* beta,
                                                    float eps, float* out, int M, int H, int K, int terms,
                                                    void* workspace, int64_t workspace_bytes, float out_scale,
                                                    void* stream);

/* Diagnostic switch (A/B timing, tools/bench_kernels.py): which kernel form serves large-M forward launches of
 * e3d_gemm_bias_act_f32_split with terms = 3 -- 4 = persistent 256x256 (default), 3 = 256x256 with interleaved
 * staging, 1 = classic 256x256 loop, 0 = 256x128, 5 = persistent whatever the tile count (experiments).  Results are identical in every form (same products, same
 * accumulation order).  pref < 0 only queries.  Returns the previous value. */
int e3d_gemm_kernel_select(int pref);
/* The same for the general kernel (every launch the persistent / 256x256 forms do not take: medium and small M, the
 * training layouts): 0 = by shape (default), 1 = 256x128 tiles (8 waves), 2 = 128x128 (4 waves), 3 = 128x128 on 8 waves,
 * 4 = 128x64 on 4 waves (48 KB of LDS: up to three workgroups per CU; forward / input-gradient layout of the 2-term
 * arithmetics, other layouts fall back to 3).  Identical results in every form (up to the order of the split-K atomics of
 * the weight-gradient layout).  form < 0 only queries. */
int e3d_gemm_general_select(int form);

/* Diagnostic (tests/test_kernels_gpu.py): the cooperative bf16x3 attention kernel keeps its running softmax maximum
 * until a key tile exceeds it by more than 2^tau (default tau = 8, log2 units; results are mathematically
 * unchanged -- O and the row sum stay relative to the same maximum).  tau = 0 restores the classic online softmax
 * (rescale on every new maximum).  tau < 0 only queries.  Returns the previous value. */
float e3d_attn_rescale_tau(float tau);

/* Process-wide switch of the split attention kernels (default 1): stop the key sweep after the tile
 * holding the last valid key of the item WHEN the call's element bounds (q_absmax / k_absmax / e_absmax of
 * e3d_relkey_attn_fwd_split_ex) prove that trailing all-padding tiles contribute exp(s - 10000 - m) = 0.0f exactly --
 * the reference masks additively (structure_model/model.py:226-231), so that holds only while the scores of a row
 * spread over less than ~9900: 16 qa (ka + ea) < 9890.  Results are then bit-identical to the dense sweep; calls
 * without bounds, or whose bounds do not prove it, always run the dense sweep.  0 = dense sweep for every call
 * (timing comparisons).  Returns the previous setting. */
int e3d_attn_skip_padded_tiles(int enable);

/* out[M,H] = LayerNorm_eps(x[M,H] (+ residual[M,H])) * gamma + beta
 * -- BertSelfOutput / BertOutput (4.38.2) with the dense bias already added by the GEMM,
 * and predictor.layer_norm (structure_model/model.py:152).  residual may be NULL.
 * s_out (may be NULL) receives the pre-norm rows x + residual that the backward needs.
 * H must be 256, 512, 768 or 1024. */
int e3d_residual_layernorm_fwd(const float* x, const float* residual, const float* gamma,
                               const float* beta, float eps, float* s_out, float* out, int M,
                               int H, void* stream);

/* SELayer gated branch (structure_model/model.py:61-67):
 *   out = x + gate * (LayerNorm_1e-5_noaffine(y) * (1 + scale) + shift)
 * mod is the adaLN_modulation output [Mc, 6H]; (shift,scale,gate) are chunks
 * (3*branch, 3*branch+1, 3*branch+2) with branch 0 = msa, 1 = mlp.  Row m of x uses row
 * m / rows_per_cond of mod (rows_per_cond = L when c is [B,1,H], 1 when c is [B,L,H]). */
int e3d_adaln_gate_fwd(const float* x, const float* y, const float* mod, int branch,
                       int rows_per_cond, float* out, int M, int H, void* stream);

/* BertEmbeddings (structure_model/model.py:111-118, eval):
 *   out[M,H] = LayerNorm_eps(x[M,F] @ W[H,F]^T + b) * gamma + beta (+ post_add[m / rows_per_add])
 * post_add ([M / rows_per_add, H], may be NULL) is the timestep embedding the sequence model
 * adds after the embedding (sequence_model/model.py:213,220).  F <= 32.  z_out (may be NULL)
 * receives the pre-LayerNorm rows x W^T + b that the backward needs. */
int e3d_embed_layernorm_fwd(const float* x, int F, const float* W, const float* b,
                            const float* gamma, const float* beta, float eps,
                            const float* post_add, int rows_per_add, float* z_out, float* out,
                            int M, int H, void* stream);

/* predictor.dense2 (structure_model/model.py:153): out[M,Nout] = x[M,H] @ W[Nout,H]^T + b,
 * Nout <= 32 (8 angles / 20 amino-acid logits). */
int e3d_head_linear_fwd(const float* x, const float* W, const float* b, float* out, int M,
                        int H, int Nout, void* stream);

/* One DDPM ancestral update (+ wrap) (structure_model/sample.py:90-99,140-142 and
 * utils.py:20-40):  mean = sqrt_recip_alpha * (x - beta * eps_hat / sqrt_one_minus_ab)
 *   v = mean + sigma * noise   (noise == NULL or sigma == 0: no noise term)
 *   out = wrap ? wrap_[-pi,pi)(v) : v          (p_sample_loop wraps, bare p_sample does not)
 * n = number of elements. */
int e3d_ddpm_step_wrap(const float* x, const float* eps_hat, const float* noise,
                       float sqrt_recip_alpha, float beta, float sqrt_one_minus_ab, float sigma,
                       int wrap, float* out, int64_t n, void* stream);

/* e3d_ddpm_step_wrap with the step index on the device: coefficients are row t_dev[0] of
 * coef_table [T,4] = (sqrt_recip_alpha, beta, sqrt_one_minus_alphas_cumprod, sigma); sigma == 0 (t == 0)
 * skips the noise term exactly like the scalar form.  No host scalar changes from step to step, so a
 * captured hipGraph of the whole reverse step can be replayed (structure_model/sample.py). */
int e3d_ddpm_step_wrap_table(const float* x, const float* eps_hat, const float* noise,
                             const float* coef_table, const int64_t* t_dev, int wrap, float* out,
                             int64_t n, void* stream);

/* Forward noising q(x_t | x_0) with wrap (structure_model/dataset.py:211-228):
 *   out[b] = wrap(sqrt_ab[t[b]] * x0[b] + sqrt_1mab[t[b]] * noise[b]),  per = elements per item. */
int e3d_q_sample_wrap(const float* x0, const float* noise, const int64_t* t,
                      const float* sqrt_ab, const float* sqrt_1mab, float* out, int B,
                      int64_t per, void* stream);

/* Discrete reverse step p(z_s | z_t) (sequence_model/sample.py:120-179), C = 20 classes:
 *   prob[n,:] = normalise( sum_x0 softmax(logits[n])[x0] * (Qt[b]^T[xt,:] * Qsb[b][x0,:]) / Qtb[b][x0,xt] )
 * with Qt = rownorm(Qsb/Qtb) computed in-kernel; x_t given as class indices [B*L] (int32; the
 * reference carries one-hots).  mode 0: argmax (diverse=False); mode 1: inverse-CDF draw with
 * uniforms u[B*L] (diverse=True).  Writes class index out_idx[B*L] and, if prob_out != NULL,
 * prob [B*L,C]. */
int e3d_discrete_posterior_sample(const int32_t* xt_idx, const float* logits, const float* Qsb,
                                  const float* Qtb, const float* u, int mode, int32_t* out_idx,
                                  float* prob_out, int B, int L, int C, void* stream);

/* Forward discrete noising (sequence_model/model.py:291-311): prob[n,:] = Qtb[b][:, x0[n]];
 * x0 index < 0 marks a padding (all-zero one-hot) row -> class 0.  u as above. */
int e3d_discrete_q_sample(const int32_t* x0_idx, const float* Qtb, const float* u, int mode,
                          int32_t* out_idx, int B, int L, int C, void* stream);

/* NeRF backbone builder, the step after structure sampling (structure_model/create_pdb.py:104-155,
 * 175-234): angles [B,L,8] fp32 in the dataset's column order
 * (phi psi omega dihedral_o tau CA:C:1N 1C:N:CA CA:C:O), lengths int32 [B] -> coords float64
 * [B,L,4,3] (N, CA, C, O per residue; residues >= length zeroed), optionally centred per pocket
 * (NERFBuilder.centered_cartesian_coords).  Bond lengths are the reference's constants. */
int e3d_nerf_backbone(const float* angles, const int32_t* lengths, double* coords, int center,
                      int B, int L, void* stream);

/* ------------------------------------------------------------------ training (backward) side
 * The reference trains through torch.autograd on these same modules (Lightning training_step,
 * structure_model/model.py:305-319, sequence_model/model.py:347-367); the entry points below are
 * the hand-written backward of each fused forward kernel. */

/* General-layout split GEMM (the three GEMMs of one nn.Linear share it):
 *   out[M,N] = act(A . B^T + bias), A logical [M,K], B logical [N,K];
 *   x_kmajor = 0: element (r,k) at X[r*ld + k];  x_kmajor = 1: at X[k*ld + r] (transposed storage).
 *   forward  y = x W^T : (0,0);  dgrad dx = dy W : A = dy (0), B = W as [K'][N'] (1);
 *   wgrad dW = dy^T x : A = dy (1), B = x (1).  K-major operands allow any K (tail zero-filled). */
int e3d_gemm_f32_split_general(const float* A, int64_t lda, int a_kmajor, const float* B, int64_t ldb,
                               int b_kmajor, const float* bias, float* out, int64_t ldc, int M, int N,
                               int K, int act, int terms, void* stream);

/* Weight and bias gradients of up to 64 linear layers of ONE shape in one launch (the backward of nn.Linear for every
 * layer of a model at once: torch.autograd computes them layer by layer, Lightning training_step as above):
 *   dW_p[N,K] (contiguous) = dz_p^T . x_p,   db_p[N] = column sums of dz_p      p = 0 .. count-1
 * dz_p [M,N] (row stride ldz), x_p [M,K] (row stride ldx); the reduction runs over the M token rows.  A single layer
 * has too few output tiles for the chip and needs split-K with atomics on a zeroed output; a model's layers together
 * fill it with whole reductions: no atomics, no memsets, deterministic.
 * dz, x, dw, db: HOST arrays of ``count`` device pointers (db may be NULL, or hold NULL entries: no bias gradient).
 * accumulate_bits: bit p set -> dW_p and db_p are added to (a weight that already holds a gradient), else overwritten.
 * Two problems of one launch must not share an output.  terms = 3 (bf16x3), 6 or 19 (bf16x6). */
int e3d_gemm_wgrad_grouped_f32_split(const float* const* dz, const float* const* x, float* const* dw, float* const* db,
                                     uint64_t accumulate_bits, int count, int64_t ldz, int64_t ldx, int N, int K, int M,
                                     int terms, void* stream);

/* The same launch for layers of DIFFERENT shapes and row strides over ONE token count M (ABI v3): problem p is
 * dW_p[N[p], K[p]] = dz_p^T x_p with row strides ldz[p] / ldx[p]; ``db`` (and single entries of it) may be NULL.  A model's
 * weight gradients then take ceil(layers / 64) launches whose grids fill the chip in whole rounds but for one tail, instead
 * of one launch per (shape, stride) group with a partial last round each. */
int e3d_gemm_wgrad_ragged_f32_split(const float* const* dz, const float* const* x, float* const* dw, float* const* db,
                                    const int* N, const int* K, const int64_t* ldz, const int64_t* ldx,
                                    uint64_t accumulate_bits, int count, int M, int terms, void* stream);

/* Backward of e3d_relkey_attn_fwd.  out / lse are the forward's outputs, dout [B,Lq,nh*64].
 * Writes dq, dk, dv (strided like q, k, v) and, with dist_emb, d_dist_emb [2P-1,64] (overwritten).
 * workspace: e3d_relkey_attn_bwd_workspace_floats(...) floats (materialised P and dS tiles + the
 * per-wave dist_emb partial blocks). */
int64_t e3d_relkey_attn_bwd_workspace_floats(int B, int nh, int Lq, int Lk, int relkey);
int e3d_relkey_attn_bwd(const float* q, int64_t q_bs, int64_t q_rs, const float* k, int64_t k_bs,
                        int64_t k_rs, const float* v, int64_t v_bs, int64_t v_rs,
                        const float* dist_emb, int P, const float* key_mask, const float* out,
                        const float* lse, const float* dout, float* dq, int64_t dq_bs, int64_t dq_rs,
                        float* dk, int64_t dk_bs, int64_t dk_rs, float* dv, int64_t dv_bs,
                        int64_t dv_rs, float* d_dist_emb, float* workspace, int B, int nh, int Lq,
                        int Lk, void* stream);

/* ---- dropout (training): reference nn.Dropout(hidden_dropout_prob) in BertEmbeddings
 * (structure_model/model.py:109-117), the SELayer MLP (model.py:45-47), BertSelfOutput / BertOutput and
 * attention_probs_dropout_prob inside BertSelfAttention (transformers 4.38.2).  Decisions are a pure
 * function of (seed, element index): 16-bit fields of a splitmix64 hash, keep iff field >= round(p*65536),
 * kept values scaled by 65536 / (65536 - round(p*65536)).  The backward pass applies the same call to the
 * gradient; nothing is stored. */
int e3d_dropout_f32(const float* x, float p, uint64_t seed, float* out, int64_t n, void* stream);

/* e3d_relkey_attn_fwd_split with dropout on the normalised probabilities (drop_p == 0: identical to it). */
int e3d_relkey_attn_fwd_split_drop(const float* q, int64_t q_bs, int64_t q_rs, const float* k, int64_t k_bs,
                                   int64_t k_rs, const float* v, int64_t v_bs, int64_t v_rs,
                                   const float* dist_emb, int P, const float* key_mask, float* out,
                                   float* lse, int B, int nh, int Lq, int Lk, int terms, float drop_p,
                                   uint64_t drop_seed, void* stream);

/* The full form of the two above: ``e_scratch`` (device, >= e3d_attn_scratch_bytes(Lk) bytes, 16-byte
 * aligned, or NULL) receives the fragment-order bf16 hi/lo planes of dist_emb that the cooperative kernel
 * reads; with NULL (and a dist_emb) the per-wave kernel serves the call instead -- the library never
 * allocates.  ``e_scratch_ready`` != 0: the scratch still holds the planes written by an earlier call with the
 * same dist_emb values and Lk (constant weights: inference) -- the 5-us pre-pass is skipped.
 * ``q_absmax`` / ``k_absmax`` / ``e_absmax`` (device, one float each, or NULL): upper bounds of |element| over the
 * call's Q rows, K rows (INCLUDING padded positions) and the distance table (required with dist_emb when the other
 * two are given) -- as left by e3d_gemm_bias_act_f32_split_ex / e3d_absmax_f32.  They only decide whether all-padding
 * key tiles may be skipped (see e3d_attn_skip_padded_tiles); NULL = never skip.  Read on the device: no host sync.
 * When the call writes the planes itself (e_scratch given, e_scratch_ready == 0) it also RAISES *e_absmax to the largest
 * |element| of the table rows it reads (round 4: the bound costs no launch of its own -- hand in a slot that holds 0 or an
 * earlier bound; a value that is already larger stays). */
int64_t e3d_attn_scratch_bytes(int Lk);
int e3d_relkey_attn_fwd_split_ex(const float* q, int64_t q_bs, int64_t q_rs, const float* k, int64_t k_bs,
                                 int64_t k_rs, const float* v, int64_t v_bs, int64_t v_rs,
                                 const float* dist_emb, int P, const float* key_mask, float* out,
                                 float* lse, int B, int nh, int Lq, int Lk, int terms, float drop_p,
                                 uint64_t drop_seed, void* e_scratch, int e_scratch_ready, const float* q_absmax,
                                 const float* k_absmax, float* e_absmax, void* stream);

/* e3d_relkey_attn_bwd for a forward that used (drop_p, drop_seed). */
int e3d_relkey_attn_bwd_drop(const float* q, int64_t q_bs, int64_t q_rs, const float* k, int64_t k_bs,
                             int64_t k_rs, const float* v, int64_t v_bs, int64_t v_rs,
                             const float* dist_emb, int P, const float* key_mask, const float* out,
                             const float* lse, const float* dout, float* dq, int64_t dq_bs,
                             int64_t dq_rs, float* dk, int64_t dk_bs, int64_t dk_rs, float* dv,
                             int64_t dv_bs, int64_t dv_rs, float* d_dist_emb, float* workspace, int B,
                             int nh, int Lq, int Lk, float drop_p, uint64_t drop_seed, void* stream);

/* The full form: ``terms`` selects the arithmetic of the backward products -- 0 = exact fp32 MFMA (what the two
 * entry points above run), 3 = bf16x3 split operands on the bf16 MFMA (fp32 accumulation; 5x fewer matrix-pipe
 * cycles), 6 = the fp32 MFMA kernels again (no bf16x6 backward: the fp32-grade mode stays exact). */
int e3d_relkey_attn_bwd_ex(const float* q, int64_t q_bs, int64_t q_rs, const float* k, int64_t k_bs,
                           int64_t k_rs, const float* v, int64_t v_bs, int64_t v_rs,
                           const float* dist_emb, int P, const float* key_mask, const float* out,
                           const float* lse, const float* dout, float* dq, int64_t dq_bs, int64_t dq_rs,
                           float* dk, int64_t dk_bs, int64_t dk_rs, float* dv, int64_t dv_bs,
                           int64_t dv_rs, float* d_dist_emb, float* workspace, int B, int nh, int Lq,
                           int Lk, int terms, float drop_p, uint64_t drop_seed, void* stream);

/* Test aid: the multipliers (0 or 1/(1-p')) the two calls above apply to P[b,h,q,key] -> out [B,nh,Lq,Lk]. */
int e3d_attn_dropout_mask(int B, int nh, int Lq, int Lk, float p, uint64_t seed, float* out, void* stream);

/* LayerNorm backward on the saved pre-norm rows s [M,H]: ds = dLN/ds, dgamma/dbeta (overwritten;
 * gamma == NULL: no affine, as SELayer.norm1/norm2). */
int e3d_layernorm_bwd(const float* dy, const float* s, const float* gamma, float eps, float* ds,
                      float* dgamma, float* dbeta, int M, int H, void* stream);

/* Backward of e3d_adaln_gate_fwd w.r.t. y (-> dy) and mod (-> dmod [Mc,6H]: stored when
 * rows_per_cond == 1, ACCUMULATED when > 1 -- zero it once before the two branch calls);
 * the gradient w.r.t. x is dout itself. */
int e3d_adaln_gate_bwd(const float* dout, const float* y, const float* mod, int branch,
                       int rows_per_cond, float* dy, float* dmod, int M, int H, void* stream);

/* Activations on saved pre-activations (training keeps z; inference fuses them into the GEMM). */
int e3d_act_fwd(const float* z, int act, float* out, int64_t n, void* stream);
int e3d_act_bwd(const float* dh, const float* z, int act, float* dz, int64_t n, void* stream);

/* out[n] = sum_m x[m,n] (bias gradients);  out[g,:] = sum of rows_per_group consecutive rows. */
int e3d_colsum(const float* x, int64_t ld, float* out, int M, int N, void* stream);
/* Up to 64 matrices of one shape transposed in one launch: dst_p[c * ld_dst + r] = src_p[r * ld_src + c], r < rows,
 * c < cols.  src, dst: HOST arrays of ``count`` device pointers.  (Training: W^T of every weight once per optimizer
 * step, for the input-gradient GEMM dz . W in the forward layout -- one launch per weight shape instead of one copy
 * kernel per layer; ld_dst lets the transposes of query / key / value land side by side as the W^T of the packed QKV.) */
int e3d_transpose_grouped_f32(const float* const* src, float* const* dst, int count, int rows, int cols, int64_t ld_src,
                              int64_t ld_dst, void* stream);
int e3d_group_sum(const float* x, int rows_per_group, float* out, int M, int H, void* stream);

/* dW[h,f] (transpose_out: [f,h]) = sum_m g[m,h] x[m,f], db[h] = sum_m g[m,h]; F <= 32
 * (BertEmbeddings.linear and predictor.dense2 weight gradients). */
int e3d_small_k_wgrad(const float* g, const float* x, float* dW, float* db, int M, int H, int F,
                      int transpose_out, void* stream);

/* dx[M,H] = dout[M,Nout] @ W[Nout,H] (input gradient of predictor.dense2). */
int e3d_head_linear_bwd_dx(const float* dout, const float* W, float* dx, int M, int H, int Nout,
                           void* stream);

/* nn.Dropout between the dense layer and the residual LayerNorm of BertSelfOutput / BertOutput (training mode; transformers
 * 4.38.2 modeling_bert.py) folded into the LayerNorm kernels: the forward applies the multipliers of e3d_dropout_f32 for
 * the same (p, seed) to x as it reads it (element index = row * H + column), the backward writes both ds (gradient of the
 * pre-norm sum = of the residual) and ds_dropped = ds * multipliers (gradient of x): no [M, H] dropout pass in either
 * direction.  Same (p, seed) in both calls. */
int e3d_residual_layernorm_drop_fwd(const float* x, const float* residual, const float* gamma, const float* beta, float eps,
                                    float* s_out, float* out, int M, int H, float drop_p, uint64_t drop_seed, void* stream);
int e3d_layernorm_bwd_drop(const float* dy, const float* s, const float* gamma, float eps, float* ds, float* ds_dropped,
                           float* dgamma, float* dbeta, int M, int H, float drop_p, uint64_t drop_seed, void* stream);
/* Both of the above in one entry point (ds_dropped == NULL: no dropout) with the parameter gradients summed through a caller
 * workspace of e3d_layernorm_bwd_workspace_floats(M, H) floats -- per-block partial rows, added up in a fixed order by a second
 * launch -- instead of float atomics on a zero-filled output: deterministic, and what autograd.layernorm_bwd calls (round 4:
 * 256-512 adders per address were most of the launch).  Backward of torch.nn.LayerNorm in BertSelfOutput / BertOutput
 * (transformers 4.38.2 modeling_bert.py) and SELayer.norm1 / norm2 (structure_model/model.py:27-67). */
int64_t e3d_layernorm_bwd_workspace_floats(int M, int H);
int e3d_layernorm_bwd_ws(const float* dy, const float* s, const float* gamma, float eps, float* ds, float* ds_dropped,
                         float* dgamma, float* dbeta, int M, int H, float drop_p, uint64_t drop_seed, float* workspace,
                         int64_t workspace_floats, void* stream);

/* ---- row-complete GEMM + bias + residual + LayerNorm (ABI v4; inference at large M) ----------------------------------
 * BertSelfOutput / BertOutput whole -- LayerNorm(dense(x) + residual), transformers 4.38.2 modeling_bert.py, reached through
 * structure_model/model.py:197-213 and sequence_model/model.py:226-231 -- as ONE launch whose workgroups own whole 768-wide
 * rows: out[M,768] = LayerNorm(A[M,K] W[768,K]^T * out_scale + bias + residual) * gamma + beta.  The pre-norm sum never
 * leaves the accumulators (the GEMM + e3d_residual_layernorm_fwd pair writes it and reads it back: 4 row passes against 3).
 * Same 2-term split products in the same order as e3d_gemm_bias_act_f32_split (terms 3 or 19: the pre-norm sums are
 * bit-identical), same two-pass statistics as e3d_residual_layernorm_fwd (outputs agree to a few 1e-7).
 *   w_planes: the weight PRE-SPLIT into its two 16-bit terms in MFMA-fragment order by e3d_weight_planes_f32_split (once
 *     per weight version; e3d_weight_planes_bytes(N, K) = N K 4 bytes; for terms = 19 split the power-of-two-scaled weight
 *     and pass the inverse power as out_scale, exactly as for the *_ex GEMM entry points);
 *   residual may be NULL; A, residual and out are row-strided (lda, ldr, ldo in floats).
 * Shapes: N = 768, M % 32 == 0, K % 32 == 0, K >= 64 (e3d_gemm_residual_layernorm_supported). */
int64_t e3d_weight_planes_bytes(int N, int K);
int e3d_weight_planes_f32_split(const float* W, int N, int K, int terms, void* planes, void* stream);
int e3d_gemm_residual_layernorm_supported(int M, int N, int K, int64_t lda);
int e3d_gemm_residual_layernorm_f32_split(const float* A, int64_t lda, const void* w_planes, const float* bias,
                                          const float* residual, int64_t ldr, const float* gamma, const float* beta, float eps,
                                          float* out, int64_t ldo, int M, int N, int K, int terms, float out_scale, void* stream);

/* ---- optimizer step (ABI v3) ------------------------------------------------------------------------------------------
 * Global-norm gradient clip + AdamW over ALL parameters in three launches: what Lightning's gradient_clip_val = 1.0
 * (structure_model/train_model.py:99-110) and torch.optim.AdamW (structure_model/model.py:361-366,
 * sequence_model/model.py configure_optimizers) do around every training_step.  Tensors are reached through pointer
 * tables in DEVICE memory (``count`` entries: params / grads / exp_avg / exp_avg_sq pointers and element counts); the work
 * unit is a chunk of e3d_optim_chunk_elems() consecutive elements of one tensor, listed by the caller in
 * ``chunk_tensor[n_chunks]`` (table index) and ``chunk_first[n_chunks]`` (first element), both device arrays.
 *   e3d_grad_global_norm: norm_and_clip[0] = sqrt(sum g^2) (deterministic: per-chunk partials into ``partial[n_chunks]``,
 *     summed in a fixed order), norm_and_clip[1] = min(max_norm / (norm + 1e-6), 1) (NaN / inf propagate, as
 *     torch.nn.utils.clip_grad_norm_ with error_if_nonfinite=False).  Gradients are NOT rewritten.
 *   e3d_adamw_step: torch's AdamW update (decoupled weight decay, bias corrections from ``step`` >= 1) with every gradient
 *     multiplied by norm_and_clip[1] on the way in (norm_and_clip may be NULL: no clip).  */
/*   e3d_adamw_step_dyn: the same update with the learning rate and the step count read from DEVICE memory
 *     (lr_and_step[0] = lr, lr_and_step[1] = number of steps taken BEFORE this one, as a float) -- the form a captured HIP
 *     graph of the training step replays: the caller advances lr_and_step[1] on the device after the launch.
 *   e3d_adamw_step_dev (ABI v4): EVERY scalar of the update from device memory -- hyper[0..5] = lr, steps taken before this
 *     one, beta1, beta2, eps, weight_decay -- so that a replayed graph follows schedulers that move more than the learning
 *     rate (torch's OneCycleLR cycles beta1 with it: structure_model/model.py:372, sequence_model/model.py:425).
 *   e3d_dropout_set_epoch_ptr: registers (per device; NULL unregisters) a device word every dropout launch adds to its
 *     seed INSIDE the kernel (e3d_dropout_f32, e3d_relkey_attn_fwd_split_drop / _ex, e3d_relkey_attn_bwd_drop / _ex,
 *     e3d_attn_dropout_mask): a replayed graph bakes the seed arguments, so the training step advances this word instead
 *     and every replay draws fresh decisions while forward and backward of one step still agree.  */
int e3d_adamw_step_dyn(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                       const int64_t* numel, const int* chunk_tensor, const int64_t* chunk_first, int n_chunks,
                       const float* norm_and_clip, const float* lr_and_step, float beta1, float beta2, float eps,
                       float weight_decay, void* stream);
int e3d_adamw_step_dev(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                       const int64_t* numel, const int* chunk_tensor, const int64_t* chunk_first, int n_chunks,
                       const float* norm_and_clip, const float* hyper, void* stream);
int e3d_dropout_set_epoch_ptr(const uint64_t* device_word);
int e3d_optim_chunk_elems(void);
int e3d_grad_global_norm(const float* const* grads, const int64_t* numel, const int* chunk_tensor, const int64_t* chunk_first,
                         int n_chunks, float max_norm, float* partial, float* norm_and_clip, void* stream);
int e3d_adamw_step(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                   const int64_t* numel, const int* chunk_tensor, const int64_t* chunk_first, int n_chunks,
                   const float* norm_and_clip, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                   void* stream);

#ifdef __cplusplus
}
#endif
#endif /* E3D_HIP_H */
