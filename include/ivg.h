/* libivg -- C ABI of the MI355X-native iVideoGPT prediction engine (gfx950 / ROCm).
 *
 * The reference (thuml/iVideoGPT) has no FFI layer: its drop-in boundary is the Python object API of
 *   CompressiveVQModel.{tokenize, detokenize, set_context_length}   ivideogpt/vq_model/compressive_vq_model.py:154-277
 *   LlamaForCausalLM.generate / HeadModelWithAction.generate       ivideogpt/transformer/action_model.py:56-121
 * (SURVEY.md 8b).  Each entry point below replaces one of those methods; the Python mirror of the
 * reference classes (the ivideogpt_amd package) binds them with ctypes, and INTEGRATION.md shows the stub a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns 0 on success, a negative ivg_status otherwise; ivg_last_error(engine) holds the text;
 *   - all data pointers are DEVICE pointers owned by the caller (e.g. torch tensor data_ptr()); the engine borrows
 *     them for the duration of the call; weight tensors passed to ivg_create must stay alive until ivg_destroy;
 *   - work is enqueued on the caller's HIP stream and is asynchronous; no host synchronisation inside, except
 *     ivg_create / ivg_destroy / ivg_profile_read and the kept-cache verification of ivg_generate_continue / ivg_generate_embeds;
 *   - an engine is bound to one device and is not thread-safe (one engine per process per GPU);
 *   - token ids are int64, pixels are float32 or bfloat16 planar (B, T, 3, H, W) in [0, 1].
 */
#ifndef IVG_H_
#define IVG_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ivg_engine ivg_engine;
typedef struct ivg_cache ivg_cache;
typedef void* ivg_stream; /* hipStream_t */

/* IVG_F32X3 (ivg_config.decode_dtype / llm_dtype only): the tensors are float32 in HBM (pass IVG_F32 pointers), the matrix products of
 * the path run in split-bf16 arithmetic -- every operand as bf16 hi + bf16 lo (2^-17), all four partial products on the bf16 MFMA
 * path, fp32 accumulation: the mode that meets the 1e-3 parity bar on pixels / logits without the f32-input MFMA rate (1/16 of bf16). */
enum ivg_dtype { IVG_F32 = 0, IVG_BF16 = 1, IVG_F32X3 = 2 };

enum ivg_status {
  IVG_OK = 0,
  IVG_ERR_INVALID = -1,   /* bad argument / shape (the reference raises AssertionError here) */
  IVG_ERR_MISSING = -2,   /* a weight tensor is missing from the table */
  IVG_ERR_HIP = -3,       /* a HIP call or kernel launch failed */
  IVG_ERR_CAPACITY = -4   /* batch / frames exceed what the engine was created for */
};

/* A named weight tensor (device pointer).  Names are the checkpoint keys of the reference
 * (SURVEY.md Appendix C); layouts are the packed ones produced by ivideogpt_amd/packing.py. */
typedef struct {
  const char* name;
  const void* data;
  int32_t dtype;
  int32_t ndim;
  int64_t shape[4];
} ivg_tensor;

typedef struct {
  /* ---- tokenizer (CompressiveVQModel.__init__ kwargs, compressive_vq_model.py:36-60); n_levels = 0: no tokenizer */
  int32_t n_levels;
  int32_t block_out_channels[8];
  int32_t layers_per_block;
  int32_t latent_channels;
  int32_t vq_embed_dim;
  int32_t num_vq_embeddings;
  int32_t num_dyn_embeddings;
  int32_t norm_num_groups;
  int32_t mid_block_add_attention;
  int32_t context_length;
  int32_t max_att_resolution;
  int32_t resolution;
  int32_t patch_size;
  /* ---- transformer (HF LlamaConfig); num_layers = 0: no transformer */
  int32_t hidden_size;
  int32_t intermediate_size;
  int32_t num_layers;
  int32_t num_heads;
  int32_t vocab_size;
  int32_t max_position_embeddings;
  float rms_norm_eps;
  int32_t action_dim;      /* 0: action-free LlamaForCausalLM; >0: HeadModelWithAction */
  int32_t reward_head;     /* 1: reward_linear present */
  /* ---- arithmetic types */
  int32_t encode_dtype;    /* tokenize path (default IVG_F32: VQ indices must match the fp32 reference; IVG_F32X3 is refused) */
  int32_t decode_dtype;    /* detokenize path: IVG_F32, IVG_BF16 or IVG_F32X3 */
  int32_t llm_dtype;       /* transformer: IVG_F32, IVG_BF16 or IVG_F32X3 */
  /* ---- capacity the workspace / KV cache are sized for */
  int32_t max_batch;       /* trajectories per call */
  int32_t max_frames;      /* frames per clip (T) */
  int32_t max_seq;         /* KV-cache length (0: max_position_embeddings) */
  /* ---- launch policy of THIS engine (no process-global state on the data path: a latency engine and a throughput engine can
   * live in one process, e.g. mbrl/video_predictor.py's step-wise rollout beside a batch evaluator) */
  int32_t decode_lds_kb;   /* LDS budget of a decode-step GEMM workgroup in KiB (16 .. 160); 0: the process default (IVG_DECODE_LDS_KB,
                            * 160 = a whole CU: fastest for one batch alone; <= 52: three or four workgroups of DIFFERENT engines share a CU --
                            * what several batches in flight on one GPU want.  A budget in force BELOW 160 -- set here or through the process
                            * default -- is the engine's BATCHES-IN-FLIGHT PROFILE: its decode GEMMs also request weights with the default
                            * cache policy -- the other engines over the same copy ask for the same lines -- and do not warm the next launch's
                            * weights; an explicit 160 is the one-batch profile, like 0 with the default untouched.  Best effort: shapes whose
                            * smallest plan is larger keep it.
                            * The budget picks the kernel generation and therefore the fp32 summation order: tokens of two budgets are
                            * each deterministic and batch-invariant but not bit-comparable with one another. */
} ivg_config;

int ivg_create(const ivg_config* cfg, const ivg_tensor* weights, int n_weights, int device, ivg_engine** out);
void ivg_destroy(ivg_engine* e);
const char* ivg_last_error(const ivg_engine* e);   /* e may be NULL: error of the last failed ivg_create */
const char* ivg_version(void);

/* The run-time switches (IVG_* environment variables, listed in ivideogpt_amd/csrc/switches.h) are read when the library is
 * loaded and at every ivg_create; a process that changes one afterwards (the A/B tests do) calls this to publish the change. */
void ivg_reload_switches(void);

/* `temperature` of every generate call of the reference (HF generate(..., temperature=...): inference/predict.py:61,
 * ivideogpt/transformer/action_model.py:61,89,104,128,143): the logits are divided by it before the top-k filter
 * (TemperatureLogitsWarper).  Engine state, default 1.0; IVG_ERR_INVALID unless strictly positive and finite (HF raises). */
int ivg_set_temperature(ivg_engine* e, float temperature);
/* ivg_config.decode_lds_kb of a live engine (0 = back to the process default); takes effect at the next generate call */
int ivg_set_decode_lds_kb(ivg_engine* e, int kb);

/* CompressiveVQModel.set_context_length (compressive_vq_model.py:154-158): keeps the LAST k frames of kv_pos_emb. */
int ivg_set_context_length(ivg_engine* e, int context_length);

/* CompressiveVQModel.tokenize (compressive_vq_model.py:164-220).
 * pixels (B, T, 3, H, W); ids_out / labels_out int64 (B, 257*ctx - 1 + 17*(T - ctx)); labels_out may be NULL. */
int ivg_tokenize(ivg_engine* e, const void* pixels, int pixel_dtype, int B, int T, int64_t* ids_out, int64_t* labels_out,
                 ivg_stream stream);

/* Context-only fast path for prediction: what predict.py:53-54 / vp/ivideogpt_interface.py:158-169 /
 * mbrl/video_predictor.py:281-283 obtain by tokenizing zero-padded clips and slicing [:, :257*ctx].
 * pixels (B, T >= ctx, 3, H, W): only the first ctx frames are read.  ids_out int64 (B, ids_stride >= 257*ctx):
 * columns [0, 257*ctx) are written (context tokens, scf separators, trailing sdf). */
int ivg_encode_context(ivg_engine* e, const void* pixels, int pixel_dtype, int B, int T, int64_t* ids_out, int64_t ids_stride,
                       ivg_stream stream);

/* CompressiveVQModel.detokenize (compressive_vq_model.py:222-277).
 * ids int64 (B, 257*ctx - 1 + 17*F); pixels_out float32 (B, ctx + F, 3, H, W), unclamped.
 * cache: NULL, or a handle from ivg_cache_create.  cache_mode 1 = fill it (return_cache=True), 2 = reuse it
 * (cache=...): context frames are then not decoded again (their pixels are copied from the cache). */
int ivg_detokenize(ivg_engine* e, const int64_t* ids, int B, int F, float* pixels_out, ivg_cache* cache, int cache_mode,
                   ivg_stream stream);
/* The same with the element type of the result chosen by the caller: pixel_dtype IVG_F32, or IVG_BF16 when the engine decodes in
 * bfloat16 (decode_dtype = IVG_BF16) -- what the reference's callers get under torch.autocast(bfloat16)
 * (vp/ivideogpt_interface.py:180, mbrl/video_predictor.py:269): half the bytes of the clip, written by the last convolution's
 * epilogue.  A cache remembers the element type it was filled with; reuse with another one is IVG_ERR_INVALID. */
int ivg_detokenize_to(ivg_engine* e, const int64_t* ids, int B, int F, void* pixels_out, int pixel_dtype, ivg_cache* cache, int cache_mode,
                      ivg_stream stream);
/* detokenize for a batch whose rows come in groups of `group_size` consecutive trajectories with the SAME context tokens (the samples of
 * one clip, predict.py:65-72; VP2's candidates, vp/ivideogpt_interface.py:184-198): ids int64 (n_groups * group_size, 257*ctx - 1 + 17*F),
 * of which columns [0, 257*ctx - 1) are read from the FIRST row of every group.  The context frames are decoded once per group (and copied
 * to its other rows), the context decoder's features and the cross-attention K / V projections of the predicted frames exist once per
 * group.  pixels_out (n_groups * group_size, ctx + F, 3, H, W) in pixel_dtype, as ivg_detokenize_to.  No cache. */
int ivg_detokenize_shared(ivg_engine* e, const int64_t* ids, int n_groups, int group_size, int F, void* pixels_out, int pixel_dtype, ivg_stream stream);
/* on != 0: ivg_detokenize writes clamp(frames, 0, 1) -- the post-processing every caller of the reference applies to the decoded
 * clip (inference/predict.py:73, vp/ivideogpt_interface.py:199, train_gpt.py:438) -- from the epilogue of the decoders' last
 * convolution instead of a separate pass over the clip.  Default off: CompressiveVQModel.detokenize returns the raw output. */
int ivg_set_output_clamp(ivg_engine* e, int on);
int ivg_cache_create(ivg_engine* e, int B, ivg_cache** out);
void ivg_cache_destroy(ivg_engine* e, ivg_cache* c);

/* LlamaForCausalLM.generate (predict.py:57-69) when actions == NULL: every new token is sampled;
 * HeadModelWithAction.generate (action_model.py:56-121) when actions != NULL: the action embedding is added to the
 * i-th sdf slot (action index i + ctx - 1) and the sdf after every 16 tokens is forced.
 *   prompt   int64 (B, L0) row stride prompt_stride, L0 = 257*ctx
 *   actions  float32 (B, act_T, action_dim) or NULL;  ctx = context length (only used with actions)
 *   uniforms float32 (B, n_new) in [0,1) or NULL (NULL = greedy argmax); column j-1 drives new token j
 *   ids_out  int64 (B, L0 + n_new): prompt followed by the new tokens
 *   reward_out float32 (B) or NULL: reward_linear(last hidden state of the final step) (mbrl/video_predictor.py:311-313) */
int ivg_generate(ivg_engine* e, const int64_t* prompt, int64_t prompt_stride, int B, int L0, int n_new, const float* actions,
                 int act_T, int ctx, const float* uniforms, int top_k, int64_t* ids_out, float* reward_out, ivg_stream stream);

/* Shared-context rollouts.  Three of the reference's four callers hand generate() rows whose prompt is ONE clip's context repeated:
 * inference/predict.py:65 (gen_input.repeat(repeat_times, 1)), train_gpt.py:165-184 (generate_multiple_times: t samples per clip),
 * vp/ivideogpt_interface.py:155-202 (VP2: every candidate action sequence starts from the same two frames).  Here the prompt is given
 * ONCE per group of `group_size` consecutive trajectories: it is prefilled once, its K / V rows are stored once, and every decode step
 * of the group's trajectories reads those rows from the one copy (L2 / Infinity Cache hits instead of group_size HBM streams); only
 * positions >= L0 - 1 -- the prompt's last token, which carries the trajectory's own action, and the new tokens -- are per trajectory.
 *   prompts  int64 (n_groups, L0) row stride prompt_stride; trajectory b = g * group_size + k uses prompts[g]
 *   actions  float32 (n_groups * group_size, act_T, action_dim) or NULL (with actions: L0 must be 257*ctx, the context alone)
 *   uniforms float32 (n_groups * group_size, n_new) or NULL (greedy);  force_sdf != 0: ivg_generate_forced_sdf's schedule
 *   ids_out  int64 (n_groups * group_size, L0 + n_new);  reward_out float32 (n_groups * group_size) or NULL
 * Same tokens as ivg_generate on the repeated prompt up to the rounding of the prompt's LAST position (fed through the decode-step
 * kernels here, through the prompt pass there): identical except at near-ties of the sampler. */
int ivg_generate_shared(ivg_engine* e, const int64_t* prompts, int64_t prompt_stride, int n_groups, int group_size, int L0, int n_new,
                        const float* actions, int act_T, int ctx, const float* uniforms, int top_k, int force_sdf, int64_t* ids_out, float* reward_out,
                        ivg_stream stream);

/* HeadModelWithAction.generate_without_action (action_model.py:123-152; no caller in the reference): 16 sampled tokens per future
 * frame, then the forced sdf separator -- ivg_generate's action-conditioned schedule without any action embedding.  Same
 * arguments as ivg_generate minus actions / reward. */
int ivg_generate_forced_sdf(ivg_engine* e, const int64_t* prompt, int64_t prompt_stride, int B, int L0, int n_new, int ctx, const float* uniforms,
                            int top_k, int64_t* ids_out, ivg_stream stream);

/* Step-wise rollout (mbrl/video_predictor.py:286-317 calls generate once per environment step on a prompt that grew by the 17
 * tokens of the previous step): same contract as ivg_generate with actions != NULL, but the engine's KV cache is taken to
 * hold positions [0, L0 - 1) of these B trajectories from the previous ivg_generate / ivg_generate_continue call, so only
 * the prompt's last token (the sdf slot that receives the new action) is fed before the 16 + 1 new tokens -- no prefill.
 * Returns IVG_ERR_INVALID when the cache does not hold exactly that: other batch, other length, cache never filled, or built from
 * other tokens / actions (the cached prefix is compared with the prompt on the device: one stream synchronisation). */
int ivg_generate_continue(ivg_engine* e, const int64_t* prompt, int64_t prompt_stride, int B, int L0, int n_new, const float* actions,
                          int act_T, int ctx, const float* uniforms, int top_k, int64_t* ids_out, float* reward_out, ivg_stream stream);

/* Embeds-level boundary of the step-wise caller (mbrl/video_predictor.py:286-317 runs these five ops per environment step).
 *
 * ivg_embed_tokens      HeadModelWithAction.get_input_embeddings (action_model.py:47-54): out[b][l][:] = embed_tokens[ids[b][l]],
 *                       out (B, L, hidden) in the engine's llm dtype.
 * ivg_action_linear     action_linear (action_model.py:36): out[r][:] = W a[r] + b, actions float32 (rows, action_dim),
 *                       out (rows, hidden) llm dtype.
 * ivg_generate_embeds   llm.generate(inputs_embeds=..., max_new_tokens, return_dict_in_generate=True, output_hidden_states=True)
 *                       (mbrl/video_predictor.py:298-313): embeds (B, L0, hidden) llm dtype; every new token is sampled (uniforms
 *                       (B, n_new) or NULL = greedy); new_ids_out int64 (B, n_new) = result.sequences (the inputs_embeds form of HF
 *                       generate returns only the new tokens); hidden_out (B, hidden) llm dtype or NULL = result.hidden_states[-1][-1],
 *                       the post-final-norm hidden state of the LAST forward pass (the one that produced the logits of new token n_new).
 *                       allow_reuse != 0: when the KV cache of the previous call on this engine was built from exactly
 *                       embeds[:, :L0 - 1] (verified on the device against a kept copy of the fed inputs -- one stream
 *                       synchronisation), only the last row is fed instead of a prefill of the grown prompt; *reused_out says which.
 * ivg_reward_linear     reward_linear (action_model.py:41; mbrl/video_predictor.py:313) on post-norm hidden rows (rows, hidden) llm
 *                       dtype -> float32 (rows). */
int ivg_embed_tokens(ivg_engine* e, const int64_t* ids, int64_t ids_stride, int B, int L, void* out, ivg_stream stream);
int ivg_action_linear(ivg_engine* e, const float* actions, int rows, void* out, ivg_stream stream);
int ivg_generate_embeds(ivg_engine* e, const void* embeds, int B, int L0, int n_new, const float* uniforms, int top_k, int64_t* new_ids_out,
                        void* hidden_out, int allow_reuse, int* reused_out, ivg_stream stream);
int ivg_reward_linear(ivg_engine* e, const void* hidden, int rows, float* out, ivg_stream stream);

/* Teacher-forced logits (LlamaForCausalLM.forward / HeadModelWithAction.forward, action_model.py:154-185):
 * ids int64 (B, L); actions as above or NULL (added on every sdf slot 257*ctx - 1 + 17*i < L); logits_out float32 (B, L, vocab). */
int ivg_logits(ivg_engine* e, const int64_t* ids, int B, int L, const float* actions, int act_T, int ctx, float* logits_out,
               ivg_stream stream);

/* Eval forward with labels (HeadModelWithAction.forward / LlamaForCausalLM.forward, action_model.py:154-205; train_gpt.py:356-376):
 * HF shifted cross-entropy, ignore_index -100, WITHOUT materialising the (B, L, vocab) fp32 logits (lm_head runs over row chunks that
 * are reduced to per-position losses in place).
 *   ids, labels    int64 (B, L); actions as in ivg_logits (added on every sdf slot) or NULL
 *   token_nll_out  float32 (B, L): -log p(labels[b][l+1] | ids[b][:l+1]) at position l; 0 where the target is ignored / l = L - 1
 *   loss_rows_out  float32 (B, 2): per trajectory (sum of token_nll, number of non-ignored targets); the HF loss of the batch is
 *                  sum(sums) / sum(counts), a trajectory's perplexity exp(sum / count)
 *   hidden_out     NULL or (B, L, hidden) llm dtype: post-final-norm hidden states (output_hidden_states=True, [-1]) -- what
 *                  reward_linear (action_model.py:198-204) and action_recon_linear (:187-196) read */
int ivg_eval_forward(ivg_engine* e, const int64_t* ids, const int64_t* labels, int B, int L, const float* actions, int act_T, int ctx,
                     float* token_nll_out, float* loss_rows_out, void* hidden_out, ivg_stream stream);
/* action reconstruction term (action_model.py:187-196): out[b] = sum over positions p >= prelude and action dims of
 * (action_recon_linear(hidden[b][p]) - actions[b][ctx - 1 + (p - prelude) / 17])^2;  mse_loss = sum(out) / (B * (L - prelude) * action_dim) */
int ivg_action_recon_sqerr(ivg_engine* e, const void* hidden, const float* actions, int B, int L, int act_T, int ctx, int prelude, float* out,
                           ivg_stream stream);

/* Clip ingest on the device: NPZParser.preprocess / EvalDataset.data_augmentation of the reference (inference/utils.py:12-16,
 * ivideogpt/data/simple_dataloader.py:512-516): frames uint8 (T, H, W, 3) as stored in the episode files -> / 255 -> optional centre
 * crop to the short side -> torchvision resize to (resolution, resolution), whose tensor path is antialiased bilinear interpolation
 * (ATen upsample_bilinear2d_aa semantics, width first) -> clip_out (T, 3, resolution, resolution) float32 or bfloat16 in [0, 1].
 * Downscale factors up to 15.  Engine-free. */
int ivg_ingest_frames(const uint8_t* frames, int T, int H, int W, int center_crop, void* clip_out, int out_dtype, int resolution,
                      ivg_stream stream);

/* Frame metrics of predicted clips on the device: Evaluator.forward of the reference without LPIPS
 * (ivideogpt/utils/video_metric.py:63-100; piqa PSNR(epsilon 1e-8, range 1) and SSIM(11 x 11 Gaussian, sigma 1.5, no padding)):
 * per frame mse / psnr / ssim, mean over the T frames of a trajectory, then the best of its t = n_samples / B samples
 * (min mse, max psnr, max ssim).  Engine-free (no weights).
 *   gt    (B, T_gt, 3, H, W) float32 or bfloat16 in [0, 1]; frames [gt_t0, gt_t0 + T) are compared
 *   pred  float32 (n_samples, T_pr, 3, H, W), frames [pr_t0, pr_t0 + T); sample k of trajectory b is row k * B + b
 *         (the layout of video_1.repeat([t, 1, 1, 1, 1]), video_metric.py:69-71)
 *   rows_out float32 (B, 3) = (mse, psnr, ssim): the per-trajectory rows the multi-GPU path all-gathers (train_gpt.py:476-479)
 *   ws    scratch of at least ivg_frame_metrics_ws_bytes(n_samples, T, H, W) bytes */
size_t ivg_frame_metrics_ws_bytes(int n_samples, int T, int H, int W);
int ivg_frame_metrics(const void* gt, int gt_dtype, int B, int T_gt, int gt_t0, const float* pred, int n_samples, int T_pr, int pr_t0, int T, int H,
                      int W, float* rows_out, void* ws, size_t ws_bytes, ivg_stream stream);

/* ---- measurement hooks (bench.py): time one kernel class with HIP events on the launching stream; the decode attention
 * (which runs inside a replayed hipGraph) stamps its own launch windows with the 100 MHz wall clock instead */
enum ivg_kernel_class { IVG_K_IGEMM_BF16 = 0, IVG_K_IGEMM_F32 = 1, IVG_K_CONV3X3_BF16 = 2, IVG_K_CONV3X3_F32 = 3, IVG_K_DECODE_ATTN = 4,
                        IVG_K_DECODE_GEMM = 5,   /* the decode step's GEMMs (q/k/v, o, gate/up, down, lm_head): HBM-bound, bytes = weights */
                        IVG_K_COUNT = 6 };
typedef struct {
  int64_t launches;
  double total_ms;      /* sum of per-launch durations (hipEventElapsedTime) */
  double total_flops;   /* algorithmic: 2 * M * N * K per launch */
  double total_bytes;   /* algorithmic: operands read once + output written once */
} ivg_profile_stats;
int ivg_profile_enable(ivg_engine* e, int kernel_class, int enable);
int ivg_profile_read(ivg_engine* e, int kernel_class, ivg_profile_stats* out);   /* synchronises, then resets */
/* after ivg_profile_read(IVG_K_DECODE_ATTN): least-squares line  launch duration = fixed_us + bytes / gbps  over the launches
 * of the last ivg_generate (their cache lengths differ) -- separates the per-launch overhead from the streaming rate */
int ivg_profile_attn_fit(ivg_engine* e, double* fixed_us, double* gbps);
/* after ivg_profile_read(IVG_K_DECODE_GEMM): mean launch window (us) and launch count of the decode-step GEMMs by kind --
 * [0] q/k/v, [1] o-proj, [2] gate/up, [3] down, [4] lm_head (arrays of 5) */
int ivg_profile_gemm_kinds(ivg_engine* e, double* mean_us, int64_t* launches);

/* ---- op-level entry points (unit parity tests call the kernels through these) */
typedef struct {
  const void* X; const void* W; void* Y; const void* R; const float* bias;
  int32_t Nimg, Hin, Win, Cin, ldx, Hout, Wout, KH, KW, stride, pad, ups, N, ldw;
  int64_t c_img, c_pix, c_ch, c_grp_stride;
  int32_t c_grp, flags;
  float alpha;
  int32_t nb0, nb1, nb2;
  int64_t sa[3], sw[3], sy[3];
} ivg_igemm_args;
/* dtype IVG_F32X3: fp32 tensors, split-bf16 arithmetic (what an x3 engine's GEMMs run: gemm256x3_kernel where it covers the shape,
 * else igemm_kernel<float, ..., X3>); flags = IG_* of csrc/igemm.h */
int ivg_op_igemm(const ivg_igemm_args* a, int dtype, ivg_stream stream);
/* 3x3 convolution whose epilogue also reduces the GroupNorm statistics of its output into gn_part (double2 [Nimg][chunks][groups],
 * at least Nimg * ceil(Hout*Wout/256) * ceil(N/64) * groups entries), then GroupNorm(+SiLU) of that output from those statistics
 * into gn_out.  Returns the number of chunks per image (> 0) or a negative ivg_status. */
int ivg_op_conv_gn(const ivg_igemm_args* a, int dtype, void* gn_part, int groups, const float* gamma, const float* beta, void* gn_out, float eps,
                   int silu, ivg_stream stream);
/* y = conv3x3(silu(GroupNorm(x))) (+ bias, residual) with the GroupNorm applied inside the convolution's input staging: the
 * normalised tensor is never written.  ws: scratch of at least Nimg * (ceil(Hin*Win/1024) * groups * 16 + Cin * 8) bytes. */
int ivg_op_gn_conv(const ivg_igemm_args* a, int dtype, int groups, const float* gamma, const float* beta, float eps, void* ws, ivg_stream stream);
/* 3x3 convolution on fp32 tensors in split-bf16 arithmetic (the "x3" decode mode): w_x3 = the [N][9 * Cin] weight matrix with every 4
 * consecutive K elements stored as [bf16 hi(4) | bf16 lo(4)] (ivideogpt_amd/packing.py: pack_x3), activations split the same way
 * inside the kernel, fp32 accumulate.  gamma != NULL: y = conv3x3(silu(GroupNorm(x))) with the normalisation inside the staging
 * (ws as ivg_op_gn_conv).  IVG_ERR_INVALID when the 3x3 kernel does not cover the shape. */
int ivg_op_conv_x3(const ivg_igemm_args* a, const void* w_x3, int groups, const float* gamma, const float* beta, float eps, void* ws,
                   ivg_stream stream);
/* Nearest-x2 upsampling followed by a 3x3 convolution (diffusers Upsample2D: vae.py:271-284) in SUB-PIXEL form: four 2x2 convolutions over
 * the low-resolution input, one per output-pixel parity, with the weights pre-summed per parity (ivideogpt_amd/packing.py pack_subpixel:
 * w_sub [4][N][4 * Cin] in the element type of X) -- 2.25 x fewer multiplies, same result.  a->ups must be 1 and a->Hout = 2 a->Hin.
 * w_x3 / w_sub_x3 (both or neither): the split-bf16 arithmetic on fp32 tensors.  gn_part / groups as in ivg_op_conv_gn (NULL: no
 * statistics).  Returns the statistics chunks per image (0 without gn_part), IVG_ERR_INVALID when the sub-pixel kernel does not cover
 * the shape. */
int ivg_op_conv_subpixel(const ivg_igemm_args* a, int dtype, const void* w_sub, const void* w_x3, const void* w_sub_x3, void* gn_part, int groups,
                         ivg_stream stream);
/* Tokenizer cross-attention in one pass (bf16 only; IVG_ERR_INVALID when the shape is not covered): q [M][P][C], Kp [M/F][kv][C],
 * VpT [M/F][C][kv] -> out [M][P][C], heads of C / nh channels, softmax(q k^T / sqrt(C / nh)) v per head
 * (ivideogpt/vq_model/conditional_vae.py:38-55). */
int ivg_op_xattn(const void* q, const void* Kp, const void* VpT, void* out, int M, int F, int P, int kv, int C, int nh, int dtype, ivg_stream stream);
/* decode-step GEMM (M <= 128 rows; K bytes a multiple of 128): Y = epi(X W^T), flags = IG_* | SK_NORM of csrc/igemm.h */
int ivg_op_skinny(const void* X, const void* W, void* Y, int M, int N, int K, int ldx, int ldw, int ldy, int flags, int dtype,
                  ivg_stream stream);
/* the same under an engine's launch policy: lds_kb = ivg_config.decode_lds_kb (0: process default), w_shared != 0: default-policy weight
 * requests -- together the batches-in-flight profile, i.e. the kernel plans bench.py's lanes run (tests compare THOSE with fp64) */
int ivg_op_skinny_policy(const void* X, const void* W, void* Y, int M, int N, int K, int ldx, int ldw, int ldy, int flags, int dtype, int lds_kb,
                         int w_shared, ivg_stream stream);
int ivg_op_groupnorm(const void* X, void* Y, void* ws /* >= N*chunks*groups*16 B */, const float* gamma, const float* beta,
                     const float* pos, int N, int P, int C, int groups, float eps, int silu, int dtype, ivg_stream stream);
int ivg_op_softmax(const float* S, void* P, int64_t rows, int Lq, int Lk, int lds, int ldp, int causal, int dtype,
                   ivg_stream stream);
int ivg_op_vq_argmin(const float* z, const float* codebook, float* ee_ws /* n_e floats */, int64_t* out, int R, int n_e,
                     ivg_stream stream);
int ivg_op_add_rmsnorm(void* x, const float* w, void* out, int M, int H, float eps, int dtype, ivg_stream stream);
int ivg_op_conv_in(const void* video, int video_dtype, const float* w, const float* bias, void* Y, int dtype, int N, int per,
                   int T_total, int t0, int H, int W, int C0, ivg_stream stream);
/* One decode-attention step of a shared-context rollout (ivg_generate_shared): qkv (B, 3 * heads * hd) of the tokens being fed (RoPE
 * at `pos` is applied inside, the new k / v are appended at cache position `pos` of every row), caches kc / vc
 * (rows, heads, Lmax, hd) in which group slot s = (b - row0) / G (row0 <= 0) holds the shared prompt rows [0, P) in cache row s and
 * every trajectory b its own rows [P, pos) in cache row b; out (B, heads * hd).  With G = 1 and P = 0: the plain step. */
int ivg_op_shared_decode_attn(const void* qkv, void* kc, void* vc, void* out, const float* cos_t, const float* sin_t, int B, int heads, int hd, int Lmax,
                              int pos, int P, int G, int row0, int dtype, ivg_stream stream);
/* one top-k draw per logits row [B][V] fp32 with the rollout's sampler (uniforms [B] in [0,1), or NULL = greedy): HF
 * TemperatureLogitsWarper (logits / temperature, > 0) + TopKLogitsWarper + softmax + draw as restated by oracle/llama.py
 * sample_from_logits */
/* The 24-bit K / V cache of the IVG_F32X3 rollout (head_dim 64): per (trajectory, head) a block of Lmax * 192 bytes -- [Lmax][64] uint16,
 * the upper halves of the fp32 values rounded to 24 bits (nearest even), then [Lmax][64] uint8, the next byte.  ivg_op_kv24_pack: fp32
 * rows [0, L) of k32 / v32 [BH][Lmax][64] -> the planes (what the prefill does per layer).  ivg_op_decode_attn24: one decode-attention
 * step at cache position pos (RoPE of q / the new k, append of the rounded k / v, softmax(q K^T / 8) V over [0, pos]); qkv [B][3 * heads * 64]
 * fp32, out [B][heads * 64] fp32; P / G / row0 as ivg_op_shared_decode_attn (G = 1: every trajectory reads its own rows). */
int ivg_op_kv24_pack(const float* k32, const float* v32, void* kc, void* vc, int BH, int L, int Lmax, ivg_stream stream);
int ivg_op_decode_attn24(const float* qkv, void* kc, void* vc, float* out, const float* cos_t, const float* sin_t, int B, int heads, int Lmax, int pos,
                         int P, int G, int row0, ivg_stream stream);
int ivg_op_sample(const float* logits, int B, int V, int top_k, float temperature, const float* uniforms, int64_t* out, ivg_stream stream);
/* test hook: launches since the library was loaded of the kernel family `name` selects ("decode_gemm_gen3" / "decode_gemm_gen2":
 * decode-step GEMMs the dispatcher sent to dgemm3.hip / dgemm.hip; "conv3x3_subpixel": upsampling convolutions run as four 2x2 phase
 * convolutions) -- lets a test assert WHICH kernel produced the tensor it checked; -1 for an unknown name */
int64_t ivg_debug_counter(const char* name);

#ifdef __cplusplus
}
#endif
#endif /* IVG_H_ */
