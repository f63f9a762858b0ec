/* myosim_ppo.h -- C ABI of the fused PPO learner kernels in libmyosim_hip.so (row f3 of DESIGN.md: the learner that consumes
 * the batched env-step, NOT part of the engine boundary of myosim.h).
 *
 * What it replaces (paths under /root/reference): the brax PPO learner the reference drives from
 *   benchmarks/mjx_benchmark_PPO.py:50-60   ppo.train(environment=env, wrap_env_fn=wrap_for_brax_training, **ppo_params)
 * with the hyper-parameters and networks of
 *   myosuite/envs/myo/mjx/__init__.py:43-67 ppo_config (policy / value MLPs of (64, 64, 64), swish, clipped surrogate,
 *                                           entropy_cost, max_grad_norm, Adam, num_minibatches x num_updates_per_batch passes).
 * brax itself is a third-party dependency of the reference (uv.lock: brax 0.14.x), absent from /root/reference; its published
 * algorithm is restated in myosuite_amd/ppo.py (torch autograd: the CHECKER for these kernels) and here as three launches per
 * minibatch update instead of ~100:
 *   mm_ppo_grad   gather + observation normalisation + policy AND value forward + clipped-surrogate / entropy / value losses +
 *                 backward (MFMA 16x16x4 f32 tiles over LDS-resident activations; 16 or 32 samples per workgroup, policy and
 *                 value networks in different workgroups of ONE launch) -> per-workgroup partial gradients, then a second
 *                 launch sums them in a fixed order into the flat gradient (deterministic: no float atomics)
 *   mm_ppo_adam   global-norm clipping + Adam on the flat parameter vector
 * and ONE launch per rollout step instead of ~30:
 *   mm_ppo_act    observation normalisation + policy forward + sampling + log-probability + squashing + value forward
 *   mm_ppo_store  reward / termination / truncation rows of the unroll buffers
 *
 * Conventions as in myosim.h: raw DEVICE pointers owned by the caller, float32 row-major, `stream` is a hipStream_t as void*,
 * nothing synchronises the host (every call can be captured into a HIP graph), 0 on success / negative MM_E* otherwise
 * (message: mm_ppo_last_error()), no CPU fallback.
 *
 * Parameter vector: policy network first, then value network; per layer the weight [out][in] row-major, then the bias [out]
 * (the order of torch.nn.Linear parameters in an nn.Sequential).  The policy's last layer has 2 act_dim outputs: the mean and
 * the raw scale (std = softplus(raw) + 1e-3); activations are swish (SiLU) on every hidden layer.
 */
#ifndef MYOSIM_PPO_H_
#define MYOSIM_PPO_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MM_PPO_MAX_LAYERS 8      /* linear layers per network (hidden + output) */
#define MM_PPO_MAX_WIDTH  128    /* widest hidden layer the fused kernels take (partial-gradient buffers and LDS are sized for it) */
#define MM_PPO_MAX_OBS    512
#define MM_PPO_MAX_OUT    256    /* 2 act_dim */

enum { MM_PPO_SQUASH_TANH = 0,       /* brax NormalTanhDistribution: action = tanh(raw) in [-1, 1] */
       MM_PPO_SQUASH_SIGMOID = 1 };  /* action = sigmoid(raw): muscle excitations in [0, 1] */

typedef struct mm_ppo mm_ppo;

typedef struct {
  int size;                       /* sizeof(mm_ppo_config) as the caller compiled it */
  int obs_dim, act_dim;
  int pi_layers, vf_layers;       /* linear layers of each network, output layer included */
  int pi_widths[MM_PPO_MAX_LAYERS];  /* output width of every layer; pi_widths[pi_layers - 1] == 2 act_dim */
  int vf_widths[MM_PPO_MAX_LAYERS];  /* vf_widths[vf_layers - 1] == 1 */
  int squash;                     /* MM_PPO_SQUASH_* */
  int max_minibatch;              /* largest minibatch mm_ppo_grad will be given (sizes the partial-gradient workspace) */
  float learning_rate, beta1, beta2, adam_eps;
  float clipping_epsilon, entropy_cost, value_cost;
  float max_grad_norm;            /* <= 0: no clipping */
} mm_ppo_config;

/* Validates the configuration and allocates the device workspace (network descriptors, partial gradients, Adam moments, step
 * counter) on `device`.  MM_EUNSUPPORTED for networks outside the MM_PPO_MAX_* limits. */
int mm_ppo_create(const mm_ppo_config* cfg, int device, mm_ppo** out);
void mm_ppo_destroy(mm_ppo* h);
const char* mm_ppo_last_error(void);
/* floats in the flat parameter (and gradient) vector */
int mm_ppo_param_count(const mm_ppo* h);
/* offset (floats) of the value network's parameters in the flat vector */
int mm_ppo_value_offset(const mm_ppo* h);
/* zero the Adam moments and the step counter (enqueued on stream) */
int mm_ppo_reset_optimizer(mm_ppo* h, void* stream);

/* One rollout step of `nenv` envs: obs [nenv][obs_dim] (the env's observation rows), obs_mean / obs_std [obs_dim] or NULL (no
 * normalisation; otherwise x = clamp((obs - mean) / std, -5, 5)), noise [nenv][act_dim] standard normal draws.  Writes
 * obs_out (a copy of the observation rows), raw_out = mean + std * noise, logp_out = log-density of the squashed action,
 * value_out, action_out = squash(raw).  action_out == NULL: value network only (the bootstrap value after the unroll);
 * obs_out / raw_out / logp_out are then ignored. */
int mm_ppo_act(mm_ppo* h, const float* params, const float* obs, const float* obs_mean, const float* obs_std, const float* noise,
               int nenv, float* obs_out, float* raw_out, float* logp_out, float* value_out, float* action_out, void* stream);

/* Unroll-buffer rows of one step: reward_out[e] = rwd[e][rwd_col] * reward_scale, trunc_out[e] = truncated[e] && ended[e],
 * term_out[e] = ended[e] && !truncated[e] (an episode that ended without hitting the time limit is a true termination). */
int mm_ppo_store(const float* rwd, int rwd_cols, int rwd_col, float reward_scale, const uint8_t* ended, const uint8_t* truncated,
                 int nenv, float* reward_out, float* trunc_out, float* term_out, void* stream);

/* Gradient of one minibatch: rows idx[0..mb) (int64) of the flattened unroll buffers obs [B][obs_dim], raw [B][act_dim],
 * logp_old / adv / ret [B].  loss = -mean(min(r A, clip(r, 1 -+ eps) A)) - entropy_cost mean(H) + value_cost mean((V - ret)^2),
 * r = exp(logp - logp_old); H = the pre-squash normal's entropy (+ the squash term when mm_ppo_set_entropy_noise gave draws).  grad_out [param_count] is overwritten; the sum of its squares is left in the workspace for
 * mm_ppo_adam.  Two launches (partials, ordered reduction). */
int mm_ppo_grad(mm_ppo* h, const float* params, const float* obs, const float* obs_mean, const float* obs_std, const int64_t* idx,
                int mb, const float* raw, const float* logp_old, const float* adv, const float* ret, float* grad_out, void* stream);

/* brax's NormalTanhDistribution.entropy = entropy of the pre-squash normal + log|d squash / d x| at a reparametrised sample
 * x = mean + std e.  `noise` = standard-normal draws e, [B][act_dim], row-indexed like `raw` (the caller refreshes them as it likes;
 * they must not depend on the parameters); NULL (the default) = pre-squash entropy only.  The pointer is kept in the handle and
 * read by every following mm_ppo_grad. */
int mm_ppo_set_entropy_noise(mm_ppo* h, const float* noise);

/* params -= Adam(clip_by_global_norm(grad * grad_scale)).  recompute_norm != 0: the gradient was modified after mm_ppo_grad
 * (data-parallel all-reduce; grad_scale = 1 / world), its norm is recomputed first (one more launch). */
int mm_ppo_adam(mm_ppo* h, float* params, const float* grad, float grad_scale, int recompute_norm, void* stream);

#ifdef __cplusplus
}
#endif
#endif
