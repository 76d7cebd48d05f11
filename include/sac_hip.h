/* sac_hip.h -- C ABI of libsac_hip.so: the MI355X-native SAC training inner loop.
 *
 * Drop-in boundary for the hot path of jeremy29tien/robosuite-benchmark:
 *   replay_buffer.random_batch(B)  ->  trainer.train(batch)
 * as driven by  /root/reference/util/rlkit_custom.py:233-240.  The reference has no FFI
 * (pure Python duck typing); each entry point below names the
 * reference interface it replaces.  Plain pointers and sizes only; no torch types.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error; sac_last_error() gives the text
 *     of the last error on the calling thread.  The library never aborts the process.
 *   - the library owns all device memory behind the opaque handles; the caller owns every
 *     host buffer passed in or out.  Host buffers are plain (pageable) memory; the library
 *     stages them through its own pinned buffers.
 *   - a handle is bound to one GPU and one HIP stream; handles are not thread-safe, distinct
 *     handles are independent.
 *   - all floating point is IEEE fp32; sampled indices are int64 (NumPy's default dtype).
 */
#ifndef SAC_HIP_H
#define SAC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sac_buffer sac_buffer_t;
typedef struct sac_trainer sac_trainer_t;

const char *sac_last_error(void);
/* number of visible GPUs (0 when there is none: every create call then fails loudly) */
int sac_device_count(void);
const char *sac_version(void);

/* ------------------------------------------------------------------------------------------
 * Replay buffer in HBM, structure-of-arrays fp32.
 * Replaces rlkit EnvReplayBuffer / SimpleReplayBuffer as constructed at
 * /root/reference/util/rlkit_utils.py:139-142 and used at
 * /root/reference/util/rlkit_custom.py:207,230 (add_paths), :235-236 (random_batch),
 * :250-253 (get_diagnostics -> 'size').
 * ------------------------------------------------------------------------------------------ */
int sac_buffer_create(sac_buffer_t **out, int64_t capacity, int obs_dim, int act_dim, int device);
int sac_buffer_destroy(sac_buffer_t *buf);

/* add_paths / add_sample: append n transitions at the ring head (top = (top+1) % capacity,
 * size = min(size+1, capacity)).  Row-major host arrays: obs (n,O) act (n,A) rew (n) next_obs (n,O)
 * term (n) uint8 (the reference's terminal dtype).  fp32 here == the reference's float64 storage
 * followed by np_to_pytorch_batch's float32 cast.  The _f64 form takes the reference's native
 * float64 arrays and rounds to fp32 on the way in. */
int sac_buffer_add(sac_buffer_t *buf, int64_t n, const float *obs, const float *act, const float *rew,
                   const float *next_obs, const uint8_t *term);
int sac_buffer_add_f64(sac_buffer_t *buf, int64_t n, const double *obs, const double *act, const double *rew,
                       const double *next_obs, const uint8_t *term);
/* The inserts are ASYNCHRONOUS (ingest overlapping env stepping; the reference inserts 2 500
 * rows per epoch between rollouts and training, /root/reference/util/rlkit_custom.py:223-231): rows are packed into
 * one of two pinned staging buffers and handed to the copy engine on the buffer's stream; the call returns once the
 * copies are enqueued, the caller's arrays are free for reuse, and size/top already count the rows.  Sampling,
 * gathering and sac_buffer_read are ordered behind the inserts on that stream, so nobody has to wait explicitly;
 * sac_buffer_ingest_pending polls (1 = rows still in flight), sac_buffer_ingest_wait blocks until they have landed. */
int sac_buffer_ingest_pending(sac_buffer_t *buf);
int sac_buffer_ingest_wait(sac_buffer_t *buf);
int64_t sac_buffer_size(const sac_buffer_t *buf);      /* 'replay_buffer/size' */
int64_t sac_buffer_top(const sac_buffer_t *buf);
int64_t sac_buffer_capacity(const sac_buffer_t *buf);

/* Checkpointing (SURVEY.md 8b "Snapshot contract": the reference saves no buffer -- rlkit's
 * get_snapshot() for it is {}, /root/reference/util/rlkit_custom.py:80 -- so a resumed run starts from an
 * empty one; here a run resumes exactly).  sac_buffer_read copies storage rows [start, start+n) to dense
 * host arrays in the sac_buffer_add layout; restore = sac_buffer_add of the saved rows in storage order,
 * then sac_buffer_set_cursor(top, size). */
int sac_buffer_read(sac_buffer_t *buf, int64_t start, int64_t n, float *obs, float *act, float *rew,
                    float *next_obs, uint8_t *term);
int sac_buffer_set_cursor(sac_buffer_t *buf, int64_t top, int64_t size);

/* The global NumPy legacy stream (np.random.seed at /root/reference/scripts/train.py:112) as the
 * replay buffer consumes it.  State layout = np.random.get_state(): key[624] + pos.  Round-tripping
 * the state keeps host NumPy consumers (env resets between training blocks) coherent. */
int sac_rng_seed(sac_buffer_t *buf, uint32_t seed);
int sac_rng_get_state(sac_buffer_t *buf, uint32_t key[624], int32_t *pos);
int sac_rng_set_state(sac_buffer_t *buf, const uint32_t key[624], int32_t pos);
/* Bind the generator to a HOST-resident MT19937 state: key[624] + *pos are the words of np.random's own state struct
 * (numpy.random.MT19937().ctypes.state_address: {uint32_t key[624]; int pos;}), so the buffer samples THE process-wide
 * stream the reference seeds at /root/reference/scripts/train.py:112 and rlkit's random_batch consumes -- by default,
 * with no per-call state transfer: every device draw is mirrored on those host words (a few ns per index), which are
 * therefore right when the call returns, and a change anybody else made to them (np.random.seed, np.random.set_state, a
 * host consumer such as an env reset) is noticed in front of the next draw and adopted.  The read-ahead of
 * sac_random_batch_device survives (the speculation is only dropped when the host words were changed from outside).
 * While bound, sac_rng_seed / sac_rng_set_state write through to the host words and sac_rng_get_state reads them.
 * key == NULL unbinds (the generator keeps its current state, privately).  The memory must outlive the binding. */
int sac_rng_bind_host(sac_buffer_t *buf, uint32_t *key, int32_t *pos);

/* np.random.randint(0, size, batch) drawn n_batches times on the device, bit-exact with NumPy
 * including rejected draws and the generator state afterwards.  idx_out (host, n_batches*batch
 * int64) may be NULL: indices then stay on the device for sac_gather_sampled / sac_train_loop. */
int sac_sample_indices(sac_buffer_t *buf, int batch, int64_t n_batches, int64_t *idx_out);

/* random_batch(batch): sample + gather + copy out.  Outputs are host arrays shaped like the
 * reference's dict entries after np_to_pytorch_batch: observations (B,O), actions (B,A),
 * rewards (B,1), terminals (B,1), next_observations (B,O), all fp32.  idx_out may be NULL. */
int sac_random_batch(sac_buffer_t *buf, int batch, float *obs, float *act, float *rew, float *term,
                     float *next_obs, int64_t *idx_out);
/* The stepwise interface without a PCIe round trip per step.  The reference's loop is
 *     train_data = replay_buffer.random_batch(bs); trainer.train(train_data)     (/root/reference/util/rlkit_custom.py:235-238)
 * where rlkit gathers on the host and copies the batch to the device inside train().  Here random_batch can leave
 * the batch where it is needed: sac_random_batch_device draws the next `batch` indices of the NumPy stream and
 * gathers them into a device slot, asynchronously, and returns a token (the batch number); sac_step_device
 * (below) runs trainer.train on it; sac_read_batch_device copies it to the host only if somebody looks at it.
 * A token stays valid until (at least) 16 more batches have been drawn (then the calls fail with "expired").
 * Read-ahead: called again and again with nothing in between that touches the generator or the buffer -- the loop above --
 * the call draws and gathers up to sixteen batches in its two launches and the following calls hand them out.  Every entry
 * point that reads or changes the generator's state, the buffer's rows or its size (add, seed / get / set state, the
 * other draws, sac_train_loop) first takes the speculation back: the generator's state saved in front of the chunk is
 * restored and advanced by the batches actually handed out.  The index stream is NumPy's, bit for bit, under any
 * interleaving (tests/test_gpu_device_batches.py).  SAC_READAHEAD=0 switches it off. */
int sac_random_batch_device(sac_buffer_t *buf, int batch, int64_t *token);
int sac_read_batch_device(sac_buffer_t *buf, int64_t token, float *obs, float *act, float *rew, float *term,
                          float *next_obs, int64_t *idx_out);

/* the gather half alone, for caller-supplied indices (parity tests, prioritised variants) */
int sac_gather(sac_buffer_t *buf, const int64_t *idx, int batch, float *obs, float *act, float *rew,
               float *term, float *next_obs);

/* Device-side batched form used for measurement: draws n_batches index vectors and gathers all
 * of them into contiguous minibatch slots in HBM in ONE gather launch; nothing is copied to the
 * host.  kernel_ms (may be NULL) receives {index kernel ms, gather kernel ms} from HIP events on
 * the handle's stream. */
int sac_sample_gather_device(sac_buffer_t *buf, int batch, int64_t n_batches, float kernel_ms[2]);
/* copy slot s of the last sac_sample_gather_device back to the host (tests) */
int sac_read_slot(sac_buffer_t *buf, int64_t slot, float *obs, float *act, float *rew, float *term,
                  float *next_obs, int64_t *idx_out);

/* ------------------------------------------------------------------------------------------
 * SAC trainer.  Replaces rlkit SACTrainer (+ TorchTrainer.train, np_to_pytorch_batch,
 * ptu.soft_update_from_to) as constructed at /root/reference/util/rlkit_utils.py:98-106 with the
 * kwargs of /root/reference/scripts/train.py:29-37 and driven at
 * /root/reference/util/rlkit_custom.py:238 (train), :258 (get_diagnostics), :63/:70 (snapshot).
 * Networks: FlattenMlp x4 (rlkit_utils.py:64-83) and TanhGaussianPolicy (:92-96).
 * ------------------------------------------------------------------------------------------ */
typedef struct sac_config {
    int32_t obs_dim, act_dim;
    int32_t hidden;              /* both hidden layers; 256 in every shipped variant.json (see policy_hidden / qf_hidden) */
    int32_t batch;               /* algorithm_kwargs.batch_size: any positive size (slots are padded to whole 16-row
                                  * blocks; pad rows carry zero weight in every mean of the step) */
    float discount;              /* trainer_kwargs.discount */
    float reward_scale;          /* trainer_kwargs.reward_scale */
    float policy_lr, qf_lr;      /* trainer_kwargs.policy_lr / qf_lr (alpha uses policy_lr) */
    float soft_target_tau;       /* trainer_kwargs.soft_target_tau */
    int32_t target_update_period;
    int32_t use_automatic_entropy_tuning;
    float target_entropy;        /* NaN => -act_dim (rlkit default) */
    uint64_t noise_seed;         /* device counter-based N(0,1) stream for rsample */
    int32_t device;
    int32_t reserved;
    /* policy_kwargs / qf_kwargs hidden_sizes (/root/reference/util/arguments.py:98,104): two layers of at most 256 units
     * each, per network family; 0 = `hidden`.  Narrower layers run EXACTLY on the 256-wide kernels: the missing units are
     * zero rows / columns whose gradients, Adam updates and Polyak averages are identically zero. */
    int32_t policy_hidden[2];
    int32_t qf_hidden[2];
} sac_config_t;

enum { SAC_NET_POLICY = 0, SAC_NET_QF1 = 1, SAC_NET_QF2 = 2, SAC_NET_TARGET_QF1 = 3, SAC_NET_TARGET_QF2 = 4,
       SAC_NET_TARGET_POLICY = 5 /* TD3 handles only */ };

/* TD3.  Replaces rlkit TD3Trainer + TanhMlpPolicy x2 as assembled at
 * /root/reference/util/rlkit_utils.py:107-135 with the trainer kwargs of /root/reference/scripts/train.py:38-47
 * (defaults /root/reference/util/arguments.py:141-156).  The handle type, the replay buffer and every sac_*
 * accessor / step / loop entry point are shared with SAC: net id 5 is the target policy, both policies are
 * [fc0, fc1, last_fc] (tanh output), sac_step's eps2 is the N(0,1) draw of the target-policy smoothing noise
 * (eps1 unused), sac_set/get_scalars carry {policy Adam steps, 0, 0, critic Adam steps, n_train_steps_total, 0}.
 * Diagnostics vector: slots 0-15 as for SAC (QF1/QF2 Loss, Policy Loss = -mean Q1(s, pi(s)), Q1/Q2 Predictions,
 * Q Targets), 16-19 Bellman Errors 1, 20-23 Bellman Errors 2, 24-27 Policy Action (Mean Std Max Min).
 * Parity: unpinned (the reference ships no TD3 run); checked against oracle/td3_step_torch.py. */
typedef struct td3_config {
    int32_t obs_dim, act_dim;
    int32_t hidden;                          /* 256 */
    int32_t batch;
    float discount;                          /* train.py:41 fixes 0.99 */
    float reward_scale;
    float policy_learning_rate, qf_learning_rate;
    float tau;                               /* soft update of all three targets, on policy steps */
    float target_policy_noise;               /* sigma of the smoothing noise (default 0.2) */
    float target_policy_noise_clip;          /* rlkit default 0.5 */
    int32_t policy_and_target_update_period; /* default 2 */
    uint64_t noise_seed;
    int32_t device;
    int32_t reserved;
    int32_t policy_hidden[2];                /* as in sac_config_t */
    int32_t qf_hidden[2];
} td3_config_t;
int td3_trainer_create(sac_trainer_t **out, const td3_config_t *cfg);
/* as sac_trainer_create_mlp: hidden_sizes of any depth / width for TD3 (two layers of at most 256 units: the fused kernels) */
int td3_trainer_create_mlp(sac_trainer_t **out, const td3_config_t *cfg, const int32_t *policy_hidden, int32_t n_policy_hidden,
                           const int32_t *qf_hidden, int32_t n_qf_hidden);

/* diagnostics vector written per step; names = the 'trainer/...' columns of progress.csv
 * (/root/reference/runs/.../progress.csv:1) plus the optimised actor loss. */
enum {
    SAC_D_QF1_LOSS = 0, SAC_D_QF2_LOSS, SAC_D_POLICY_LOSS /* logged: mean(log_pi - q_new) */,
    SAC_D_ACTOR_LOSS /* optimised: mean(alpha*log_pi - q_new) */,
    SAC_D_Q1_MEAN, SAC_D_Q1_STD, SAC_D_Q1_MAX, SAC_D_Q1_MIN,
    SAC_D_Q2_MEAN, SAC_D_Q2_STD, SAC_D_Q2_MAX, SAC_D_Q2_MIN,
    SAC_D_QT_MEAN, SAC_D_QT_STD, SAC_D_QT_MAX, SAC_D_QT_MIN,
    SAC_D_LOGPI_MEAN, SAC_D_LOGPI_STD, SAC_D_LOGPI_MAX, SAC_D_LOGPI_MIN,
    SAC_D_MU_MEAN, SAC_D_MU_STD, SAC_D_MU_MAX, SAC_D_MU_MIN,
    SAC_D_LOGSTD_MEAN, SAC_D_LOGSTD_STD, SAC_D_LOGSTD_MAX, SAC_D_LOGSTD_MIN,
    SAC_D_ALPHA, SAC_D_ALPHA_LOSS,
    SAC_DIAG_N = 32
};

int sac_trainer_create(sac_trainer_t **out, const sac_config_t *cfg);
/* hidden_sizes of ANY depth (variant['policy_kwargs'|'qf_kwargs']['hidden_sizes'], /root/reference/util/arguments.py:98,104;
 * FlattenMlp / TanhGaussianPolicy take a list of any length: /root/reference/util/rlkit_utils.py:64-97): 1..7 hidden layers
 * of 1..4096 units per network family (cfg->hidden / policy_hidden / qf_hidden are ignored).  Two hidden layers of at most
 * 256 units -- every shipped variant.json -- run on the fused kernels exactly as with sac_trainer_create; every other shape
 * runs the GENERAL step: the same step in the same order as a sequence of 2 Lp + 2 Lq + 2 launches around one
 * matrix-product kernel (csrc/sac_general.h), results within fp32 round-off of the oracle like the fused kernels'
 * (tests/test_gpu_general_shapes.py); sac_trainer_step_kind reports 3.  Every sac_* entry point works on such a handle except
 * sac_profile_loop.  (TD3: td3_trainer_create_mlp.) */
int sac_trainer_create_mlp(sac_trainer_t **out, const sac_config_t *cfg, const int32_t *policy_hidden, int32_t n_policy_hidden,
                           const int32_t *qf_hidden, int32_t n_qf_hidden);
int sac_trainer_destroy(sac_trainer_t *t);

/* flat fp32 parameter vector of one net, nn.Linear layout (W (out,in) row-major, then b):
 *   Q nets : fc0.W fc0.b fc1.W fc1.b last_fc.W last_fc.b
 *   policy : fc0.W fc0.b fc1.W fc1.b last_fc.W last_fc.b last_fc_log_std.W last_fc_log_std.b */
int64_t sac_param_count(const sac_trainer_t *t, int net);
int sac_set_params(sac_trainer_t *t, int net, const float *flat, int64_t n);
int sac_get_params(sac_trainer_t *t, int net, float *flat, int64_t n);
/* Adam state of the three trained nets (same flat layout) + step count; the reference's snapshot
 * omits these (SURVEY.md section 5 "Checkpoint / resume"), true resume needs them. */
int sac_set_opt_state(sac_trainer_t *t, int net, const float *exp_avg, const float *exp_avg_sq, int64_t n);
int sac_get_opt_state(sac_trainer_t *t, int net, float *exp_avg, float *exp_avg_sq, int64_t n);
/* scalars[6] = {log_alpha, alpha_exp_avg, alpha_exp_avg_sq, adam_step_count, n_train_steps_total, alpha} */
int sac_set_scalars(sac_trainer_t *t, const double scalars[6]);
int sac_get_scalars(sac_trainer_t *t, double scalars[6]);

/* trainer.train(np_batch) == np_to_pytorch_batch + train_from_torch: ONE gradient step on a
 * caller-supplied batch (host fp32, shapes as sac_random_batch writes them).  eps1/eps2 (B,A) are
 * the N(0,1) draws of the two rsample() calls (policy on obs, then on next_obs); NULL => the
 * device counter-based stream.  diag (SAC_DIAG_N floats, may be NULL) receives this step's stats. */
int sac_step(sac_trainer_t *t, const float *obs, const float *act, const float *rew, const float *term,
             const float *next_obs, const float *eps1, const float *eps2, float *diag);

/* trainer.train(batch) on a device-resident batch (token of sac_random_batch_device on `buf`): four kernel
 * launches behind the gather, no host copy, no synchronisation -- unless diag != NULL, which copies the step's
 * diagnostics out (rlkit reads them on the first step of an epoch only).  Noise: the device stream. */
int sac_step_device(sac_trainer_t *t, sac_buffer_t *buf, int64_t token, float diag[SAC_DIAG_N]);

/* The whole hot loop of rlkit_custom.py:234-238 on the device:
 *   for _ in range(n_steps): batch = buffer.random_batch(B); trainer.train(batch)
 * Indices for all n_steps are drawn first (same stream consumption as n_steps random_batch calls:
 * nothing else touches np.random or the buffer inside the loop), gathered into HBM slots, then the
 * steps run back to back with no host round trip.  diag_first / diag_last (may be NULL) receive the
 * diagnostics of the first and last step (the reference logs the first step of each epoch). */
/* Calls in a row: a call that directly follows another sac_train_loop on the same buffer (nothing in between that touches
 * the generator, the rows or the slots) leaves the NEXT call's first four batches drawn and gathered behind its own, under
 * its last steps, so that the next call's first step has no draw and no gather in front of it (~15 us of a call).  That
 * chunk is speculation: the generator's state as everybody sees it (sac_rng_get_state, np.random when bound) stays behind
 * the batches handed out, and any other entry point that comes first takes the speculation back.  Results and index
 * stream are those of calls that never speculate, bit for bit (tests/test_gpu_train_loop.py). */
int sac_train_loop(sac_trainer_t *t, sac_buffer_t *buf, int64_t n_steps, float *diag_first, float *diag_last);

/* 1 while the trainer runs the FUSED step: launches A + B + C of the step as one launch (k_abc) whose workgroups hand
 * their partial sums to each other inside the launch, then the weight-gradient / Adam launch -- two launches per step
 * instead of four.  It needs the chip to itself (every workgroup resident).  A hand-off that times out (50 ms) makes
 * the step apply NOTHING, the call that notices returns an error, and the trainer falls back to the four-launch step
 * for good; SAC_FUSED=0 in the environment selects the four-launch step from the start.  Both give identical results
 * (tests/test_gpu_fused_step.py). */
int sac_trainer_is_fused(const sac_trainer_t *t);
/* How a trainer's step is launched: 0 = four launches (k_fwd_a, k_fwd_b, k_bwd, k_dw_adam); 1 = the fused step above
 * (k_abc, k_dw_adam: batches up to 256 rows); 2 = three launches for batches whose 256-wide layers are not split over
 * workgroups (1024 rows and more): the two forward launches as ONE, k_chain8 -- a workgroup runs the policy, takes its own
 * head and goes on into the Q nets, no hand-off involved -- then k_bwd and k_dw_adam; 3 = the general step of
 * sac_trainer_create_mlp (network shapes beyond the fused kernels'); 4 = kind 2 with the backward blocks inside the chained
 * launch behind in-launch hand-offs (k_chain8<..., BWD>, then k_dw_adam: two launches; chosen when the launch's 4 x
 * row-blocks workgroups fill the CUs exactly -- batch 1024 on 256 CUs -- and the first layers are one k-chunk;
 * SAC_CHAIN_BWD=0/1 overrides).  Like kind 1 it can give up inside a launch; the trainer then re-runs that step as kind 2
 * and stays there (sac_trainer_step_kind reports 2 from then on), results unchanged.  (TD3: 0 or 1.) */
int sac_trainer_step_kind(const sac_trainer_t *t);

/* measurement helpers: HIP events on the trainer's stream around the last sac_train_loop (total_ms == steps_ms: first
 * launch to last step), and around the index draw (sample_ms) and the gather (gather_ms) of ONE of its chunks -- the
 * second chunk when the call has one (12 batches at the default plan), else the first. */
int sac_sync(sac_trainer_t *t);
int sac_last_loop_ms(sac_trainer_t *t, float *total_ms, float *sample_ms, float *gather_ms, float *steps_ms);

/* profiling pass: the same loop with HIP events (on the launching streams) around every kernel.
 * out_ms[9] = {index kernel, gather kernel, then the MEAN per-launch ms of k_fwd_a, k_fwd_b, k_bwd,
 * a reserved slot (0), k_dw_adam (fused step: k_abc in the k_fwd_a slot, 0 in the next two) (each minus the cost of an empty event pair), that empty-pair cost,
 * and the wall ms of all n_steps steps}.  n_steps <= 4096. */
int sac_profile_loop(sac_trainer_t *t, sac_buffer_t *buf, int64_t n_steps, float out_ms[9]);

/* Experiment hooks (bench.py --replicas-per-gpu R --xcd-replicas): confine a handle's launches to
 * the CUs of one XCD (0..7) through a CU-masked stream, so that eight independent runs -- the reference's real workload is
 * many independent jobs, /root/reference/launch_jobs.sh:15-24 -- can share one GPU without each launch spanning the chip.
 * A trainer confined this way takes the four-launch step (the fused step needs all its workgroups resident).
 * The _mask variants take a set of XCDs (bit k = XCD k): a trainer whose fused step fits the CUs it is given (batch 128
 * on four XCDs) KEEPS the fused step and is no longer serialised with the other trainers of the process -- the caller
 * promises that trainers confined this way own disjoint XCDs; mask 0xff undoes the confinement. */
int sac_buffer_set_xcd(sac_buffer_t *buf, int xcd);
int sac_trainer_set_xcd(sac_trainer_t *t, int xcd);
int sac_buffer_set_xcd_mask(sac_buffer_t *buf, unsigned xcd_mask);
int sac_trainer_set_xcd_mask(sac_trainer_t *t, unsigned xcd_mask);

/* SURVEY.md 8d "Bounding roofline": the peaks the roofline fractions divide by, MEASURED on the box -- a float4
 * stream copy of 1 GiB (read + write GB/s) and a back-to-back v_mfma_f32_16x16x4_f32 loop on every SIMD (TFLOP/s).
 * out[4] = {copy GB/s, fp32 MFMA TFLOP/s, GB moved per copy pass, ms of the best MFMA pass}. */
int sac_measure_peaks(int device, float out[4]);

/* test access to intermediates of the last step: name in {"a_new","log_pi","mu","log_std","q1","q2",
 * "q_target","q1_new","q2_new","a_next","log_pi_next","g_policy","g_qf1","g_qf2"} (g_* = flat
 * gradient in the sac_get_params layout).  Returns the element count, <0 on error. */
int64_t sac_debug_fetch(sac_trainer_t *t, const char *name, float *out, int64_t cap);

/* policy.get_action(obs) on the HOST with weights mirrored from the device (acting path,
 * /root/reference/util/rlkit_custom.py:437; MakeDeterministic => tanh(mean)).
 * eps (A floats) may be NULL when deterministic. */
int sac_policy_mirror(sac_trainer_t *t);
int sac_policy_act(sac_trainer_t *t, const float *obs, int deterministic, const float *eps, float *act);

#ifdef __cplusplus
}
#endif
#endif /* SAC_HIP_H */
