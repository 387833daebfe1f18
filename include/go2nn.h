/* go2nn.h — C ABI of the policy-side MFMA kernels (libgo2nn_hip.so, gfx950).
 *
 * What it replaces, in the rollout of OnPolicyRunner.learn (rsl_rl/rsl_rl/runners/on_policy_runner.py:135-153): PPO.act
 * (rsl_rl/rsl_rl/algorithms/ppo.py:90-102) = ActorCritic.act + evaluate + get_actions_log_prob
 * (rsl_rl/rsl_rl/modules/actor_critic.py:119-136): two 4-layer MLPs (Linear, ELU, ..., Linear; :50-75), a Gaussian sample
 * a = mu + std * eps, its log-probability, and the rows PPO.act keeps for RolloutStorage.add_transitions
 * (rsl_rl/rsl_rl/storage/rollout_storage.py:88-101).  In PyTorch that is ~30 launches per env step (8 GEMMs, 6 ELUs, the sampling head)
 * at M = 4096 rows — launch-latency-bound; here it is ONE launch: a workgroup carries 32 rows through all layers of one network on the
 * matrix pipe, activations in LDS, weights streamed from L2 in a pre-packed operand order.  Arithmetic: fp32 operands split EXACTLY into
 * three bf16 planes, six v_mfma_f32_32x32x16_bf16 terms per product, fp32 accumulate (csrc/go2nn_mlp3.h; no operand bit is dropped, the
 * results are as close to float64 as an fp32 evaluation's); with GO2_GEMM_SPLIT=0 in the environment, or for networks whose activations
 * do not fit the LDS as planes (two neighbouring 512-wide layers), v_mfma_f32_32x32x2_f32 on the fp32 values (csrc/go2nn_impl.cpp).
 *
 * Plain pointers and sizes, no torch types; asynchronous on the given HIP stream; 0 = ok, negative = error (go2nn_last_error).
 * All pointers are device pointers unless stated. */
#ifndef GO2NN_H
#define GO2NN_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GO2NN_ABI_VERSION 6      /* 2: + the learner-side kernels (go2nn_head_backward, go2nn_linear_*); 3: + the grouped (actor + critic) layer calls; 4: + split-operand (3 x bf16) products;
                                    5: + the CTS pieces: layers without activation, plain input gradients, the latent normaliser forward / backward, the split surrogate, two-segment policy inputs;
                                    6: + Go2nnBwdInJob.x_in: the weight gradient of the layer below out of the input gradient's epilogue (its gz_prev never goes to HBM) */
#define GO2NN_MAX_LAYERS 6
#define GO2NN_MAX_WIDTH 512      /* widest layer input / output (LDS holds two 32-row activation tiles of this width) */
#define GO2NN_EINVAL (-22)
#define GO2NN_EDEVICE (-5)

/* An MLP as torch.nn.Sequential(Linear, ELU(alpha=1), ..., Linear) holds it: weight[l] is [dims[l+1], dims[l]] row-major (out x in),
 * bias[l] is [dims[l+1]].  Host struct with device pointers. */
typedef struct Go2nnMlp {
  int32_t num_layers;
  int32_t dims[GO2NN_MAX_LAYERS + 1];
  const float* weight[GO2NN_MAX_LAYERS];
  const float* bias[GO2NN_MAX_LAYERS];
} Go2nnMlp;

int go2nn_abi_version(void);
const char* go2nn_last_error(void);

/* Number of floats of the packed operand buffer of `m` (weights in MFMA B-operand order for both arithmetics — fp32 zero-padded to 32 x 8 tiles, the three bf16 planes
 * to 32 x 16 —, + padded biases; the layout is the library's own);
 * negative on an unsupported shape (more than GO2NN_MAX_LAYERS layers, a dimension above GO2NN_MAX_WIDTH). */
int64_t go2nn_packed_floats(const Go2nnMlp* m);
/* Which kernel evaluates `m` in the calls below (a host-side query, no device work): 3 = the split-operand kernel (csrc/go2nn_mlp3.h: three bf16 planes, six bf16-MFMA terms),
 * 1 = the fp32-MFMA kernel (GO2_GEMM_SPLIT=0 in the environment, or activations that do not fit the LDS as planes), 0 = the host test build's loops; negative on an unsupported
 * shape.  bench.py and the GPU tests read it so that a fall-back to the slower kernel cannot go unnoticed. */
int32_t go2nn_mlp_arith(const Go2nnMlp* m);
/* Re-pack the CURRENT weights of `m` into `packed` (call after every optimizer step that the next forward must see; one small launch). */
int go2nn_pack(const Go2nnMlp* m, float* packed, void* stream);

/* y[N, dims[last]] = m(x[N, dims[0]]) — one launch (ActorCritic.act_inference / evaluate, actor_critic.py:131-136). */
int go2nn_mlp_forward(const Go2nnMlp* m, const float* packed, const float* x, float* y, int32_t N, void* stream);

/* PPO.act for one policy step (ppo.py:90-102), one launch:
 *   mu = actor(obs), value = critic(critic_obs), a = mu + std * eps (two separately rounded operations, like the eager formulation),
 *   log_prob = sum_j [-(a_j - mu_j)^2 / (2 std_j^2) - log std_j - log sqrt(2 pi)]   (j ascending)
 * a_out [N,A] receives the actions (the tensor env.step gets); the *_st pointers receive the storage rows of this step and may be NULL:
 * actions [N,A], mu [N,A], sigma [N,A], log-prob [N], value [N].  eps [N,A] are standard-normal draws supplied by the caller. */
int go2nn_policy_act(const Go2nnMlp* actor, const float* actor_packed, const Go2nnMlp* critic, const float* critic_packed,
                     const float* obs, const float* critic_obs, const float* std, const float* eps,
                     float* a_out, float* a_st, float* mu_st, float* sig_st, float* lp_st, float* v_st, int32_t N, void* stream);

/* ---- ABI 5: the same kernel in the rollout of the Concurrent Teacher-Student runner (rsl_rl/rsl_rl/runners/on_policy_runner_cts.py:135-160 -> algorithms/cts.py:112-149 ->
 * modules/actor_critic_cts.py:146-176): per step the reference gathers teacher / student env rows, runs teacher_encoder(privileged obs) resp. student_encoder(history),
 * normalises, cats [latent | obs] and [latent | privileged obs], runs actor and critic, samples — ~45 launches.  Here: TWO launches.
 *   go2nn_mlp_forward_rows   up to two MLPs in one launch, each on its own row subset `rows[0 .. nrows)` of its inputs (the teacher / student envs; NULL: rows 0 .. nrows - 1),
 *                            its input row made of two column segments (x: columns [0, kx), x2: the rest), its output stored at the SAME row index of y (pitch ldy),
 *                            L2-normalised (F.normalize: x / max(|x|, 1e-12)) when `normalize` — both encoders write the env-ordered latent [N, L] directly
 *   go2nn_policy_act_latent  go2nn_policy_act with the actor's input = [latent | obs] and the critic's = [latent | critic_obs] read as two segments (no cat);
 *                            actor->dims[0] = L + obs width, critic->dims[0] = L + critic_obs width */
typedef struct Go2nnMlpIO {
  const float* x; const float* x2;      /* input segments (x2 may be NULL when kx = the network's input width) */
  const int32_t* rows;                  /* device int32 [nrows] or NULL */
  float* y;
  int32_t ldx, ldx2, kx, nrows, ldy, normalize;
} Go2nnMlpIO;
int go2nn_mlp_forward_rows(const Go2nnMlp* const* nets, const float* const* packed, const Go2nnMlpIO* io, int32_t nnets, void* stream);
int go2nn_policy_act_latent(const Go2nnMlp* actor, const float* actor_packed, const Go2nnMlp* critic, const float* critic_packed,
                            const float* latent, int32_t L, const float* obs, const float* critic_obs, const float* std, const float* eps,
                            float* a_out, float* a_st, float* mu_st, float* sig_st, float* lp_st, float* v_st, int32_t N, void* stream);

/* ---- learner side: PPO.update's backward pass (rsl_rl/rsl_rl/algorithms/ppo.py:120-187; autograd over modules/actor_critic.py:50-75) ----
 *
 * Backward of an MLP's tail  h -> Linear -> ELU(alpha=1) -> y [B,K] -> Linear(W [C,K], b [C]) -> out [B,C]  for a NARROW output (C <= 16: the
 * 12-wide action mean, the 1-wide value), given gy = dLoss/d out [B,C], in one streaming pass over y:
 *   gz [B,K]  = (gy W) * (y > 0 ? 1 : y + 1)      the gradient at the hidden layer's PRE-activation (torch: mm + elu_backward(is_result))
 *   sums      = [ dW [C,K] = gy^T y | gb [K] = column sums of gz (the hidden Linear's bias gradient) | db [C] = column sums of gy ]
 * i.e. what autograd computes with two GEMMs of degenerate shape, a split-K fix-up, two column-sum reductions and an element-wise pass.
 * Sums are formed in a fixed order (per-workgroup partials, then a fixed tree): bit-reproducible from run to run.
 * K a multiple of 4, K <= GO2NN_MAX_WIDTH.  workspace: go2nn_head_backward_workspace(B, C, K) floats (negative: unsupported shape). */
/* Second stage of the fixed-order reductions: out[c] = sum_r part[r][c] (part [nrows, ncols] row-major) for up to 16 jobs in ONE launch.
 * go2nn_head_backward (sums == NULL) and go2nn_linear_backward_input (gb_prev == NULL) then leave their per-workgroup partial rows in `workspace` —
 * go2nn_*_rows rows of (C + 1) K + C resp. Kin columns — and the caller finishes all of a backward pass's reductions (and the row splits of its
 * weight gradients) with one go2nn_sum_rows call, off the chain of dependent GEMMs. */
typedef struct Go2nnSumJob { const float* part; float* out; int32_t nrows, ncols; float* acc; int32_t nacc, pad_; int32_t out_w, out_ld; } Go2nnSumJob;      /* ABI 4: acc != NULL: acc[c] += out[c] for c < nacc (a running sum over launches, e.g. the update's mean losses) */
/* ABI 6: out_w > 0: the sums are a [ncols / out_w, out_w] matrix written with row pitch out_ld — a column block of a wider matrix (a weight gradient assembled from
 * two launches' partials: Go2nnBwdInJob.x_in, Go2nnBwdWJob.ldx); 0: dense */
#define GO2NN_MAX_SUM_JOBS 32      /* (ABI 5: was 16 — a CTS policy step finishes 19 reductions in one launch) */
int go2nn_sum_rows(const Go2nnSumJob* jobs, int32_t njobs, void* stream);
int32_t go2nn_head_backward_rows(int32_t B, int32_t C, int32_t K);
int32_t go2nn_linear_backward_input_rows(int32_t M, int32_t C, int32_t Kin);
int64_t go2nn_head_backward_workspace(int32_t B, int32_t C, int32_t K);
int go2nn_head_backward(const float* gy, const float* y, const float* w, float* gz, float* sums, float* workspace, int32_t B, int32_t C, int32_t K, void* stream);

/* The hidden layers  y = elu(x W^T + b)  of the same MLPs, forward and backward, as fp32-MFMA GEMMs whose epilogues do the element-wise work that
 * follows a vendor GEMM as separate passes over the [M, N] activations (torch: addmm + elu_; mm + elu_backward + sum(0); a row-split bmm + sum(0)).
 * x [M,K], W [N,K] (torch.nn.Linear layout), b [N], y [M,N]; all row-major and dense.  Any M, N, K >= 1 (ragged edges are masked; 16-byte loads
 * are used when a row length is a multiple of 4).  ELU through the hardware exponential as in go2nn_policy_act. */
int go2nn_linear_elu_forward(const float* x, const float* w, const float* b, float* y, int32_t M, int32_t K, int32_t N, void* stream);
/* Backward of a layer with C outputs and Kin inputs, given gz [M,C] = the gradient at ITS pre-activation:
 *   go2nn_linear_backward_input:  gz_prev [M,Kin] = (gz W) * (y_prev > 0 ? 1 : y_prev + 1)   where y_prev [M,Kin] = the layer's input = the previous
 *                                 layer's ELU output; gb_prev [Kin] = column sums of gz_prev (the previous layer's bias gradient)
 *   go2nn_linear_backward_weight: dw [C,Kin] = gz^T x                                          (x [M,Kin] = the layer's input)
 * both with fixed-order sums (bit-reproducible).  workspace: go2nn_linear_backward_workspace(M, C, Kin) floats serve either call. */
int64_t go2nn_linear_backward_workspace(int32_t M, int32_t C, int32_t Kin);
int go2nn_linear_backward_input(const float* gz, const float* w, const float* y_prev, float* gz_prev, float* gb_prev, float* workspace,
                                int32_t M, int32_t C, int32_t Kin, void* stream);
int go2nn_linear_backward_weight(const float* gz, const float* x, float* dw, float* workspace, int32_t M, int32_t C, int32_t Kin, void* stream);


/* ---- ABI 3: one layer of SEVERAL independent MLPs per launch (PPO.update evaluates the same layer of the actor and of the critic back to back:
 * ppo.py:131-133 -> actor_critic.py:119-136; as two launches on two HIP streams the pair takes twice one network's time and needs the second stream).
 * A group is 1..GO2NN_MAX_GROUP jobs; the jobs' tiles form one grid.  Same arithmetic and the same fixed summation orders as the single calls.
 *   forward:       y = elu(x W^T + b) per job (M, K may differ between the jobs; the 45- and 263-wide input layers are one group)
 *   input grad:    gz_prev = (gz W) * elu'(y_prev); column partial sums of gz_prev are left in the job's `workspace` as
 *                  go2nn_linear_backward_input_group_rows(M, C, Kin) rows of Kin columns (the caller finishes them with go2nn_sum_rows)
 *   weight grad:   row-slice partials of dW = gz^T x are left in the job's `workspace` as go2nn_linear_backward_weight_group_rows(jobs, njobs) rows of
 *                  C * Kin columns (every job of a group has the same M); operands go from global memory straight into MFMA registers (both are
 *                  contiguous along the output index), the four waves of a workgroup split the rows of one output tile
 * workspace floats per job: rows * Kin resp. rows * C * Kin. */
#define GO2NN_MAX_GROUP 2
/* ABI 5 (what was padding; 0 keeps the ABI 3 / 4 meaning).  Every job of a group carries the same flag.
 *   Go2nnFwdJob.act     0: y = elu(x W^T + b);  1: y = x W^T + b — the LAST Linear of an encoder, whose output goes to a normaliser instead of an ELU
 *                       (rsl_rl/rsl_rl/modules/actor_critic_cts.py:49-80: teacher_encoder / student_encoder = MLP -> L2Norm)
 *   Go2nnBwdInJob.plain 0: gz_prev = (gz W) * elu'(y_prev) + column partials;  1: gz_prev = gz W — the gradient at the INPUT of a network's first layer
 *                       (CTS: d loss / d [latent | obs], actor_critic_cts.py:146-151 -> autograd); y_prev and workspace are not read
 *   Go2nnBwdInJob.ld    row pitch in floats of y_prev and gz_prev, 0 = Kin (dense).  A pitch > Kin addresses a column block of a wider matrix: the E expert heads of
 *                       a MoE encoder (rsl_rl/rsl_rl/modules/utils.py:78-93, Conv1d(groups=E)) read their 128 columns of the shared [M, E * 128] backbone output and
 *                       write the matching block of its gradient, ELU' and bias partials in the epilogue, with no transposing copy in between; the job's
 *                       workspace stays dense (rows x Kin) */
typedef struct Go2nnFwdJob { const float *x, *w, *b; float* y; int32_t M, K, N; int32_t act; const void* w_split; } Go2nnFwdJob;
 /* ABI 6 (appended fields; NULL / 0 keep the ABI 5 meaning).  Every job of a group alike; split-operand kernels only (w_split set), plain 0, ld 0.
 *   Go2nnBwdInJob.x_in  the input x [M, Kx] (dense, 1 <= Kx <= 64) of the layer BELOW — the one whose ELU output is y_prev.  The job then ALSO leaves that layer's weight
 *                       gradient  dW_prev [Kin, Kx] = gz_prev^T x_in  as go2nn_linear_backward_input_fused_rows(M) partial rows of Kin * Kx columns in `dw_workspace`
 *                       (finished by go2nn_sum_rows), formed from the gz_prev tile while it is still in the accumulators: autograd's mm(gz_prev.t(), x) (ppo.py:173
 *                       loss.backward() through the FIRST Linear of actor_critic.py:50-75) without gz_prev's round trip through HBM.  gz_prev may then be NULL (PPO: nothing
 *                       else reads the gradient at the first layer's pre-activation); `workspace` keeps its meaning and its row count.
 *   Go2nnBwdInJob.ldx   row pitch of x_in in floats, 0 = Kx: x_in may be a column block of a wider input — a 263-wide critic input leaves its last 7 columns' weight
 *                       gradient here and its first 256 = two whole 128-column tiles to go2nn_linear_backward_weight_group (Go2nnBwdWJob.ldx), which would otherwise pad
 *                       263 to 384 columns */
typedef struct Go2nnBwdInJob { const float *gz, *w, *y_prev; float *gz_prev, *workspace; int32_t M, C, Kin; int32_t plain; const void* w_split; int32_t ld;
                               int32_t Kx; const float* x_in; float* dw_workspace; int32_t ldx; } Go2nnBwdInJob;
typedef struct Go2nnBwdWJob { const float *gz, *x; float* workspace; int32_t M, C, Kin; int32_t split; int32_t ldx; } Go2nnBwdWJob;      /* ABI 6: ldx = row pitch of x, 0 = Kin (split-operand kernel only) */
int go2nn_linear_elu_forward_group(const Go2nnFwdJob* jobs, int32_t njobs, void* stream);
int32_t go2nn_linear_backward_input_group_rows(int32_t M, int32_t C, int32_t Kin);
int go2nn_linear_backward_input_group(const Go2nnBwdInJob* jobs, int32_t njobs, void* stream);
int32_t go2nn_linear_backward_input_fused_rows(int32_t M);          /* ABI 6: partial rows of dw_workspace */
int32_t go2nn_linear_backward_weight_group_rows(const Go2nnBwdWJob* jobs, int32_t njobs);
int go2nn_linear_backward_weight_group(const Go2nnBwdWJob* jobs, int32_t njobs, void* stream);

/* ---- ABI 4: the same three products on the bf16 matrix pipe WITHOUT giving up fp32 operands.  Every fp32 value is split exactly into three bf16 planes
 * (8 + 8 + 8 significand bits) and a product is six bf16 MFMA terms accumulated in fp32 — the three terms left out are below 2^-23 of the product, i.e. below
 * the rounding of an fp32 multiply; against float64 the results are as close as the fp32-MFMA kernels' (tools/gemm3_bench.cpp, tests/test_gpu_mlp_tail.py hold both
 * to the same tolerances; rms error 0.8 - 1.2 x).  One measurable difference: the bf16 pipe's accumulation is not round-to-nearest-even, every output carries a bias of
 * about a third of an fp32 ulp towards -inf, which a column sum over 24576 rows turns into ~1.5e-6 of the sum (fp32 kernels: ~4e-7).  v_mfma_f32_32x32x16_bf16 moves 16 k per 32 cycles, v_mfma_f32_32x32x2_f32 2 k per 64: six terms cost 3/8 of the fp32 form.
 *   go2nn_split_weights   a layer's weight [N,K] -> its split image for both orientations (forward: rows n; input gradient: rows k), `go2nn_split_weights_bytes`
 *                         bytes in a caller-owned buffer; run once after every optimizer step (one launch for up to 8 layers)
 *   Go2nnFwdJob.w_split / Go2nnBwdInJob.w_split   that image: non-NULL selects the split-operand kernel for the group (every job of a group alike)
 *   Go2nnBwdWJob.split    1 selects it for the weight gradient (both operands are activations: split in registers, no image)
 * NULL / 0 keep the fp32-MFMA kernels (bit-for-bit the ABI 3 results). */
typedef struct Go2nnSplitJob { const float* w; void* image; int32_t N, K; } Go2nnSplitJob;
#define GO2NN_MAX_SPLIT_JOBS 16      /* (ABI 5: was 8 — teacher encoder + actor + critic of a CTS policy step are 9 hidden layers) */
int64_t go2nn_split_weights_bytes(int32_t N, int32_t K);
int go2nn_split_weights(const Go2nnSplitJob* jobs, int32_t njobs, void* stream);


/* The narrow heads of BOTH networks forward, the PPO loss head (rsl_rl/rsl_rl/algorithms/ppo.py:131-170: Gaussian log-prob, ratio, clipped surrogate, clipped
 * value loss, entropy, KL) with its analytic gradients, and the heads backward, as ONE streaming pass over the last hidden activations y_a, y_c [B,K] of the
 * actor (head W_mu [A,K], b_mu [A]) and the critic (head w_v [1,K], b_v [1]) — in place of two degenerate GEMMs, go2sim_ppo_loss and two go2nn_head_backward
 * launches.  Same arithmetic as go2sim_ppo_loss (torch.max tie splitting, clamp gradient semantics) with every row weighted 1 / B.
 * Out: gz_a, gz_c [B,K] = the gradients at the last hidden layers' pre-activations; `partials` = go2nn_ppo_heads_rows(B, A, K) rows of
 * go2nn_ppo_heads_cols(A, K) columns whose column sums (go2nn_sum_rows) are
 *   [ surrogate, value loss, KL, entropy (means) | d loss/d std [A] | dW_mu [A,K] | gb_a [K] | db_mu [A] | dW_v [K] | gb_c [K] | db_v ]
 * (gb_*: bias gradient of the last hidden layer = column sums of gz_*).  A <= 16, K a multiple of 4 up to 256.  Fixed summation order.
 * ABI 5: surrogate_split = n > 0 gives the CTS surrogate (rsl_rl/rsl_rl/algorithms/cts.py:228-231): mean over rows [0, n) + mean over rows [n, B) — a row's surrogate
 * and its gradient are weighted 1 / n resp. 1 / (B - n) instead of 1 / B (go2sim_ppo_loss's surrogate_split); value loss, KL and entropy keep 1 / B. */
typedef struct Go2nnPpoHeads {
  const float *y_a, *y_c, *w_mu, *b_mu, *w_v, *b_v, *std, *actions, *old_mu, *old_sigma, *old_logp, *adv, *old_values, *returns;
  float *gz_a, *gz_c, *partials;
  int32_t B, A, K, use_clipped_value_loss;
  float clip, value_loss_coef, entropy_coef;
  int32_t surrogate_split;
} Go2nnPpoHeads;
int32_t go2nn_ppo_heads_rows(int32_t B, int32_t A, int32_t K);
int32_t go2nn_ppo_heads_cols(int32_t A, int32_t K);
int go2nn_ppo_heads(const Go2nnPpoHeads* a, void* stream);


/* ---- ABI 5: the latent normaliser of the Concurrent Teacher-Student networks (rsl_rl/rsl_rl/modules/utils.py:24-30 L2Norm = F.normalize(x, p=2, dim=-1): x / max(|x|, 1e-12);
 * rsl_rl/rsl_rl/modules/actor_critic_cts.py:49-80,146-176: latent = L2Norm(encoder MLP), actor input = [latent | obs], critic input = [latent.detach() | privileged obs]).
 * Row-wise streaming kernels over [n, L] (L a multiple of 4, <= 128); fixed summation orders.
 *
 * go2nn_latent_concat: zhat = z / max(|z|, 1e-12) written into columns [0, L) of up to two row-major destinations with their own row pitch (the [latent | obs] and
 *   [latent | privileged obs] input matrices of the actor and the critic: torch.cat + two copies in the reference), inv_norm [n] = 1 / max(|z|, 1e-12) kept for the backward pass
 *   (dst_b, inv_norm may be NULL).
 * go2nn_l2norm_backward: g = d loss / d zhat read from columns [0, L) of a matrix with row pitch ldg (the plain input gradient of the actor's first layer), zhat likewise
 *   (pitch ldz):  dz = (g - zhat (zhat . g)) * inv_norm  -> dz [n, L] dense; column partial sums of dz (the encoder's last bias gradient) as
 *   go2nn_l2norm_backward_rows(n) rows of L columns in `partials` (finished by go2nn_sum_rows).
 * go2nn_latent_mse: the student step of CTS (rsl_rl/rsl_rl/algorithms/cts.py:259-275): shat = normalise(z_s), that = normalise(z_t),
 *   loss = mean((that - shat)^2) over n L elements; dz_s = gradient of the loss at the student encoder's un-normalised output z_s.  `partials` = go2nn_l2norm_backward_rows(n)
 *   rows of 4 + L columns: [ loss, 0, 0, 0 | column sums of dz_s ].  grad_scale multiplies dz_s (1: the plain MSE). */
int32_t go2nn_l2norm_backward_rows(int32_t n);
int go2nn_latent_concat(const float* z, int32_t n, int32_t L, float* dst_a, int32_t lda, float* dst_b, int32_t ldb, float* inv_norm, void* stream);
int go2nn_l2norm_backward(const float* g, int32_t ldg, const float* zhat, int32_t ldz, const float* inv_norm, float* dz, float* partials, int32_t n, int32_t L, void* stream);
int go2nn_latent_mse(const float* z_s, const float* z_t, float* dz_s, float* partials, int32_t n, int32_t L, float grad_scale, void* stream);

/* The loss head of the MoE student step (rsl_rl/rsl_rl/modules/utils.py:96-152: MoE.forward + StudentMoEEncoder's normaliser; rsl_rl/rsl_rl/algorithms/moe_cts.py:203-214):
 *   w = softmax(logits [n, E]);  y = sum_e w_e outs[:, e, :] (outs [n, E, L]);  shat = y / max(|y|, 1e-12);  latent loss = mean((t_hat - shat)^2)   (t_hat [n, L]: the teacher's
 *   NORMALISED latent);  usage = mean over rows of w;  load balance = mean_e((usage_e - 1 / E)^2);  loss = latent + lb_coef * load balance
 * with its analytic gradients — two launches (the load-balance gradient needs the batch mean of the gate first) in place of ~55 element-wise / reduction launches of autograd:
 *   go2nn_moe_usage      partials = go2nn_l2norm_backward_rows(n) rows of E: column partial sums of w (go2nn_sum_rows -> usage_sum [E], sums not means)
 *   go2nn_moe_mix_loss   d_logits [n, E], d_outs [n, E, L] = d loss / d logits, / d outs;  partials = go2nn_l2norm_backward_rows(n) rows of 4: [ latent loss | load balance (row 0) | 0 | 0 ]
 * E <= 16; L as above.  Fixed summation order. */
int go2nn_moe_usage(const float* logits, float* partials, int32_t n, int32_t E, void* stream);
int go2nn_moe_mix_loss(const float* logits, const float* outs, const float* t_hat, const float* usage_sum, float* d_logits, float* d_outs, float* partials,
                       int32_t n, int32_t E, int32_t L, float lb_coef, int32_t expert_major, const float* bias, float* dbias_partials, void* stream);
/* ABI 6: the mixture FORWARD only — the student rows of a rollout step (rsl_rl/rsl_rl/algorithms/cts.py:112-149: CTS.act evaluates the student encoder without gradient):
 *   z[rows ? rows[r] : r][0 .. L) = normalise(sum_e softmax(logits[r])_e (outs[r, e] + bias[e]))     z has row pitch ldz (a multiple of 4, 16-byte aligned rows)
 * one launch in place of softmax, a broadcast product, a sum, the normaliser's four element-wise / reduction kernels and the index_copy into the env-ordered latent. */
int go2nn_moe_mix_forward(const float* logits, const float* outs, const float* bias, const int32_t* rows, float* z, int32_t ldz, int32_t n, int32_t E, int32_t L,
                          int32_t expert_major, void* stream);
/* expert_major 1: outs / d_outs are [E, n, L] (the batched GEMM's own layout: no transposing copies either way).  bias (optional, [E, L]): the expert heads' output bias,
 * added to outs here — autograd then differentiates a plain batched product — with its gradient left as go2nn_l2norm_backward_rows(n) partial rows of E L columns in
 * dbias_partials (a 77 us torch reduction otherwise). */

#ifdef __cplusplus
}
#endif
#endif
