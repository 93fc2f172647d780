/*
 * rp_engine.h -- C ABI of the MI355X-native self-play bin-packing engine
 * (librp_engine.so, built from resource_packing_self_play_amd/csrc/rp_engine.hip).
 *
 * The reference (Wang-Xiaoyang/resource_packing_self_play) is pure Python and has
 * no FFI; its boundary for this path is the duck-typed plugin API that
 * xw_mcts/main_bpp.py:93-121 and xw_mcts/CoachBPP.py drive.  Each entry point
 * below names the reference method(s) it replaces (paths relative to
 * /root/reference/xw_mcts).  The Python classes with the reference's names
 * (resource_packing_self_play_amd.{binpacking.BinPackingGame, MCTS_bpp, CoachBPP,
 * binpacking.pytorch.NNet}) are thin ctypes callers of this ABI; INTEGRATION.md
 * shows the binding a reference maintainer would add.
 *
 * Conventions
 *   - plain C types only; no torch / HIP types in signatures (streams and device
 *     buffers cross as void* / raw device pointers).
 *   - every function returns 0 on success or a negative rp_status; the message is
 *     available from rp_last_error().  No exception crosses the ABI.
 *   - the caller owns every buffer it passes; the library owns everything reachable
 *     from rp_ctx.  Inputs are never modified unless documented as in/out.
 *   - one rp_ctx per (process, GPU), externally synchronised; all device work is
 *     enqueued on the stream given at create time.
 *   - a state is (rows, remaining): rows[r] is the W-bit occupancy of grid row r
 *     (bit c = cell (r, c)), always passed as uint64; remaining[i] != 0 iff item i is
 *     still unplaced; item sizes are (w, h) bytes.  action = item * W + column, as
 *     BinPackingGame.py:67,91.
 *   - limits: 1 <= W <= 64, 1 <= H <= 64, 1 <= N <= 128.
 */
#ifndef RP_ENGINE_H
#define RP_ENGINE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define RP_ABI_VERSION 4  /* 4: sparse replay buffer (rp_config.max_sparse, rp_examples_packed*, rp_expand_examples), rp_leaf_count_async */

typedef enum rp_status {
    RP_OK = 0,
    RP_ERR_ARG = -1,        /* bad argument */
    RP_ERR_DEVICE = -2,     /* HIP error (no GPU, launch failure, ...) */
    RP_ERR_CAPACITY = -3,   /* a per-game node / edge / table arena overflowed */
    RP_ERR_ASSERT = -4,     /* a precondition the reference asserts was violated
                               (BinPackingGame.py:69 item already placed, :89 no legal move) */
    RP_ERR_STATE = -5       /* call not valid in the engine's current phase */
} rp_status;

/* how a finished search turns into a move (CoachBPP.py:86-87 draws from OS entropy,
 * which cannot be reproduced; the engine offers deterministic rules instead) */
typedef enum rp_move_rule {
    RP_MOVE_EXTERNAL = 0,   /* host supplies actions through rp_advance_roots */
    RP_MOVE_ARGMAX_FIRST = 1, /* lowest-index argmax of the root visit counts */
    RP_MOVE_SAMPLE = 2      /* a ~ counts with a counter-based RNG keyed by (seed, episode id, move) */
} rp_move_rule;

/* game phases reported by rp_game_status */
enum { RP_PHASE_IDLE = 0, RP_PHASE_RUNNING = 1, RP_PHASE_WAIT_EVAL = 2, RP_PHASE_MOVE_READY = 3,
       RP_PHASE_EPISODE_DONE = 4, RP_PHASE_FAILED = 5 };

/* kinds of a stored Q / backed-up value (NumPy promotion state):
 * 0 weak (Python int/float, f64 arithmetic), 1 float32, 2 strong float64 */
enum { RP_KIND_WEAK = 0, RP_KIND_F32 = 1, RP_KIND_F64 = 2 };

typedef struct rp_config {
    int32_t abi_version;   /* RP_ABI_VERSION */
    int32_t W, H, N;       /* BinPackingGame(bin_width, bin_height, num_items, n) BinPackingGame.py:15 */
    int32_t games;         /* concurrent game slots G on this GPU */
    int32_t sims;          /* args.numMCTSSims (MCTS_bpp.py:37) */
    double cpuct;          /* args.cpuct (MCTS_bpp.py:114,117); must be > 0 (see rp_commit_eval) */
    double alpha;          /* args.alpha (MCTS_bpp.py:79) */
    int32_t node_cap;      /* per-game node arena; 0 = sims * (N + 1) + 2 */
    int32_t edge_cap;      /* per-game legal-move arena (one 6-byte entry per legal move of every node); 0 = automatic */
    int32_t move_rule;     /* rp_move_rule */
    int32_t auto_restart;  /* 1: a finished slot pulls the next instance from the pool set by rp_set_instance_pool */
    int32_t reclaim;       /* 1: when a move is played, the arena chunks of the old root's level are recycled (its states
                              are unreachable from then on); do not re-root to shallower states afterwards */
    int32_t reserved0;
    uint64_t seed;         /* RNG seed for RP_MOVE_SAMPLE and the tie rule */
    uint64_t tie_salt;     /* salt of the deterministic stand-in for np.random.choice([1,-1]) (BinPackingGame.py:212) */
    int32_t device;        /* HIP device ordinal */
    int32_t vis_cap;       /* per-game visited-edge arena (32-byte entries); 0 = automatic */
    void *stream;          /* hipStream_t; NULL = the default stream */
    int64_t max_examples;  /* capacity of the replay buffer in examples; 0 = none recorded */
    int64_t max_sparse;    /* capacity of the replay buffer's pool of (action, visit count) pairs -- one per visited root edge of every
                              recorded example; 0 = automatic (max_examples x min(A, sims + 1, 64)) */
} rp_config;

typedef struct rp_ctx rp_ctx;

/* ---- life cycle ---------------------------------------------------------------------- */
int rp_version(void);
/* MCTS.__init__ (MCTS_bpp.py:16-26) + BinPackingGame.__init__ (BinPackingGame.py:15-22) for G games */
int rp_create(const rp_config *cfg, rp_ctx **out);
void rp_destroy(rp_ctx *ctx);
/* message of the last failing call on this ctx (or of rp_create when ctx == NULL) */
const char *rp_last_error(const rp_ctx *ctx);
/* bytes of HBM the ctx holds */
int64_t rp_device_bytes(const rp_ctx *ctx);

/* ---- stateless game rules over a batch of B states (host pointers) -------------------- */
/* BinPackingGame.getValidMoves (BinPackingGame.py:78-92) -> Bin.get_moves_for_square /
 * get_adjacency (BinPackingLogic.py:47-93).  mask_out[b][a] in {0,1}; n_valid_out may be NULL.
 * A state without a legal move yields an all-zero row (the shim raises the reference's
 * AssertionError of BinPackingGame.py:89). */
int rp_valid_moves(rp_ctx *ctx, int64_t B, const uint64_t *rows /*[B][H]*/, const uint8_t *remaining /*[B][N]*/,
                   const uint8_t *item_wh /*[B][N][2]*/, uint8_t *mask_out /*[B][W*N]*/, int32_t *n_valid_out /*[B]*/);
/* BinPackingGame.getNextState (BinPackingGame.py:58-76) -> Bin.execute_move (BinPackingLogic.py:95-109).
 * status_out[b] = 0, or RP_ERR_ASSERT when the item is already placed (BinPackingGame.py:69). */
int rp_apply_move(rp_ctx *ctx, int64_t B, const uint64_t *rows, const uint8_t *remaining, const uint8_t *item_wh,
                  const int32_t *action /*[B]*/, uint64_t *rows_out, uint8_t *remaining_out, int32_t *status_out);
/* BinPackingGame.getGameEnded (BinPackingGame.py:109-116) -> has_valid_moves (:94-107) and
 * getRankedReward (:188-212).  ended_out[b] = 0 (a move exists), +1, -1, or 2 for the r == bl
 * tie branch (the caller draws); reward_out[b] = r.  rewards is the R2 buffer shared by the batch. */
int rp_game_ended(rp_ctx *ctx, int64_t B, const uint64_t *rows, const uint8_t *remaining, const uint8_t *item_wh,
                  const int32_t *total_area /*[B]*/, const int32_t *max_h /*[B]*/, const double *rewards, int32_t n_rewards,
                  double alpha, int32_t *ended_out, double *reward_out);

/* ---- episode set-up ------------------------------------------------------------------ */
/* Items of game slot(s): BinPackingGame.getInitItems (BinPackingGame.py:37-51, also fixes max_h)
 * and CoachBPP.items_total_area (CoachBPP.py:34,119).  Starts a new episode for slots
 * first..first+count-1 at the empty board with an empty tree (CoachBPP.py:67-68,124). */
int rp_begin_episodes(rp_ctx *ctx, int32_t first, int32_t count, const uint8_t *item_wh /*[count][N][2]*/,
                      const int32_t *total_area /*[count]*/, const uint64_t *episode_id /*[count] or NULL*/);
/* Pool of instances for auto_restart: slot g takes instance g at rp_begin_pool and a finished
 * slot takes the next unused one; episode id = pool index + first_id. */
int rp_set_instance_pool(rp_ctx *ctx, int64_t n_instances, const uint8_t *item_wh /*[n][N][2]*/,
                         const int32_t *total_area /*[n]*/, uint64_t first_id);
int rp_begin_pool(rp_ctx *ctx);
/* ItemsGenerator.items_generator(seed) (BinPackingGame.py:257-285) on device: guillotine cuts of the bin_w x bin_h rectangle
 * into N items, bit-identical to the reference's NumPy stream (np.random.seed = MT19937 init_genrand, legacy randint = masked
 * rejection on 32-bit draws).  rp_generate_items returns the (w, h) pairs to the host; rp_set_instance_pool_seeds fills the
 * auto_restart pool directly on device (total area bin_w * bin_h as CoachBPP.py:119). */
int rp_generate_items(rp_ctx *ctx, int64_t n, const uint32_t *seeds /*[n]*/, int32_t bin_w, int32_t bin_h, uint8_t *item_wh_out /*[n][N][2]*/);
int rp_set_instance_pool_seeds(rp_ctx *ctx, int64_t n_instances, const uint32_t *seeds, int32_t bin_w, int32_t bin_h, uint64_t first_id);
/* R2 buffer snapshot (rewards_list argument of MCTS.getActionProb / getGameEnded, CoachBPP.py:76-91):
 * the threshold sorted[int(floor(len*alpha))-1] is recomputed and used by every slot from now on. */
int rp_set_rank_buffer(rp_ctx *ctx, const double *rewards, int32_t n);
/* Re-root slot(s) at an arbitrary state without clearing the tree (MCTS.getActionProb called with a
 * new canonicalBoard on the same MCTS object, CoachBPP.py:74-78). */
int rp_set_roots(rp_ctx *ctx, int32_t first, int32_t count, const uint64_t *rows /*[count][H]*/,
                 const uint8_t *remaining /*[count][N]*/);

/* Moves the context to another HIP stream (e.g. the side stream of a hipGraph capture: rp_search_step(ctx, NULL),
 * rp_leaf_planes and rp_commit_eval only enqueue work, so a whole simulation wave can be captured and replayed). */
int rp_set_stream(rp_ctx *ctx, void *stream);
/* Bounds the simulations one slot runs inside one rp_search_step (0 = unbounded, the default).  Slots near the end of a
 * game hit cached terminal states (Es, MCTS_bpp.py:81-83) and need no evaluator; the cap keeps such a slot from stretching
 * the launch for everyone.  Scheduling only -- results are identical for any value. */
int rp_set_step_cap(rp_ctx *ctx, int32_t max_sims_per_step);
/* Compact evaluator rows without a host round trip: with enable != 0, rp_search_step(ctx, NULL) also lists the waiting slots in
 * slot order on the device (row b = b-th waiting slot, as with a count request), rp_leaf_stem / rp_leaf_planes / rp_commit_eval
 * follow that list, and the rp_nn_resstage16 / rp_nn_resstage32 / rp_nn_convpool32 launches of this context stop at the number of
 * waiting leaves (read on the device), so slots that wait for nothing -- finished episodes, terminal streaks -- cost no
 * convolution work.  Rows past the count hold stale data.  Scheduling only: results do not depend on it. */
int rp_set_compact_rows(rp_ctx *ctx, int32_t enable);
/* Switches the move rule; onehot_examples != 0 records pi as a one-hot on the played action, the greedy branch of
 * MCTS.getActionProb (greedy_a == 0, MCTS_bpp.py:43-49) that CoachBPP uses after iterStepThreshold (CoachBPP.py:132). */
int rp_set_move_rule(rp_ctx *ctx, int32_t move_rule, int32_t onehot_examples);
/* Changes args.numMCTSSims for the following searches (MCTS_bpp.py:37); rp_search_step stops a slot after this many
 * simulations from its current root. */
int rp_set_sims(rp_ctx *ctx, int32_t sims);

/* ---- search: MCTS.search (MCTS_bpp.py:56-139) in lock step over all slots ------------- */
/* Runs simulations for every RUNNING slot until each either needs a leaf evaluated (the
 * nnet.predict call of MCTS_bpp.py:87) or has finished its args.numMCTSSims budget
 * (select :106-121, descend :125-128, terminal :78-83, backup :130-139 on device).
 * Enqueues work only; *n_leaves_out (may be NULL) forces a stream sync and returns the number of
 * leaves waiting for evaluation. */
int rp_search_step(rp_ctx *ctx, int32_t *n_leaves_out);
/* Writes the evaluator input of the waiting leaves, FP32 NCHW [n][N+1][H][W] exactly as
 * BinPackingGame.getBinItem + NNet.predict's view (BinPackingGame.py:118-120, NNet.py:77-79),
 * into caller-owned DEVICE memory (capacity_rows rows). */
int rp_leaf_planes(rp_ctx *ctx, float *planes_dev, int64_t capacity_rows);
/* Measurement aid: enqueues a copy of the current number of waiting leaves (the device-side count of rp_set_compact_rows /
 * rp_search_step) into HOST memory (pinned, or the copy synchronises) without waiting for it. */
int rp_leaf_count_async(rp_ctx *ctx, int32_t *count_host);
/* Evaluator stem on device: the network's first convolution + max-pool (BinpackingNNet.py:34,39-40:
 * conv_seqs[0].conv 3x3 pad 1, N+1 -> 16 channels, then max_pool2d(3, stride 2, pad 1)) evaluated straight from the packed
 * leaf states, exploiting that item planes are origin-anchored rectangles (tabulated tap sums).  rp_stem_set_weights reads
 * the DEVICE weight [16][N+1][3][3] and bias [16] tensors and rebuilds the tables (call again after every weight update);
 * rp_leaf_stem writes FP32 [n][16][(H+1)/2][(W+1)/2] into caller-owned DEVICE memory, the input of conv_seqs[0].res_block0.
 * The tap sums are tabulated in per-channel fixed point (float64 sums quantised to int32, unit 2^-k chosen from the channel's
 * bound), added as integers and converted to float32 once: within (N + 2) half units + half an ulp of the exact sum, i.e. the
 * correctly rounded convolution for practical purposes; rp_leaf_planes + the two PyTorch ops differ from it by their own
 * float32 summation error. */
int rp_stem_set_weights(rp_ctx *ctx, const float *conv_w_dev, const float *bias_dev);
int rp_leaf_stem(rp_ctx *ctx, float *out_dev, float *out_relu_dev /* relu(out), may be NULL */, int64_t capacity_rows,
                 int32_t channels_last /* 0: [n][16][Hp][Wp], 1: [n][Hp][Wp][16] (MIOpen's FP32 kernels are faster on NHWC) */);
/* Fused element-wise pieces of the evaluator on the context's stream (NCHW float32 DEVICE tensors).  PyTorch-ROCm runs the
 * bias add of a convolution, each ReLU, the residual add and the max-pool of BinpackingNNet.py:21-27,39-40 as separate
 * HBM-bound kernels; these apply the same operations in the same order in one pass:
 *   rp_nn_bias_relu      x = relu(x + bias[c])                                       (in place)
 *   rp_nn_bias_residual  out = (x + bias[c]) + res ; out_relu = relu(out)            (out_relu may be NULL)
 *   rp_nn_bias_pool      out = max_pool2d(x + bias[c], 3, stride 2, pad 1) ; out_relu = relu(out)
 * Channels-last tensors: call the first two with (B*H*W, C, 1), the pool with channels_last = 1. */
/* Fused residual block of the 16-channel stage on the FP32 matrix cores, channels-last [B][H][W][16]:
 *   out = x + conv1(relu(conv0(relu(x)) + b0)) + b1 ; out_relu = relu(out)   (BinpackingNNet.py:21-27)
 * rp_nn_pack_conv16 reorders a contiguous [16][16][3][3] weight into the kernels' fragment order, [9 taps][64 lanes][4]:
 * frag[tap][lane][j] = W[co = lane & 15][ci = 4 * (lane >> 4) + j][tap] (a lane's four k-steps of a tap are four consecutive input
 * channels: one 16-byte load). */
int rp_nn_pack_conv16(rp_ctx *ctx, const float *w_dev, float *frag_dev);
/* Value head in one pass: out[b] = tanh(dot(z[b, :K], w) + bias[0])  (value_fc + tanh, BinpackingNNet.py:70,81); K a multiple of 4. */
int rp_nn_value_head(rp_ctx *ctx, const float *z_dev, const float *w_dev, const float *bias_dev, float *out_dev, int64_t B, int32_t K);
int rp_nn_resblock16(rp_ctx *ctx, const float *x_dev, const float *frag0_dev, const float *bias0_dev, const float *frag1_dev, const float *bias1_dev,
                     float *out_dev, float *out_relu_dev /* may be NULL */, int64_t B, int32_t H, int32_t W);
/* Both residual blocks of a 16-channel stage (ConvSequence.res_block0 then res_block1, BinpackingNNet.py:41-46) in one launch for
 * images of at most 128 pixels: frag4 = four rp_nn_pack_conv16 fragments (16-byte aligned) and bias4 = [4][16] in execution order
 * (block 0 conv0, conv1, block 1 conv0, conv1).  out_relu_dev may be NULL.  The product is taken transposed (weights x pixels), so
 * x, the skip operands and the result move as 16-byte pieces of the channels-last tensors; persistent waves. */
int rp_nn_resstage16(rp_ctx *ctx, const float *x_dev, const float *frag4_dev, const float *bias4_dev, float *out_dev, float *out_relu_dev, int64_t B,
                     int32_t H, int32_t W);
/* The same for a 32-channel stage on images of at most 80 pixels (5x5 at the 20x20 board).  rp_nn_pack_conv32 reorders a
 * contiguous [32][Cin][3][3] weight (Cin 16 or 32) into [9 taps][Cin / 16 halves][2 M tiles][64 lanes][4]:
 * frag[tap][h][mt][lane][j] = W[co = 16 mt + (lane & 15)][ci = Cin / 4 * (lane >> 4) + 4 h + j][tap];
 * frag4 = four of those (Cin 32, 16-byte aligned), bias4 = [4][32], in execution order. */
int rp_nn_pack_conv32(rp_ctx *ctx, const float *w_dev, float *frag_dev, int32_t Cin);
int rp_nn_resstage32(rp_ctx *ctx, const float *x_dev, const float *frag4_dev, const float *bias4_dev, float *out_dev, float *out_relu_dev, int64_t B,
                     int32_t H, int32_t W);
/* First convolution of a 32-channel stage + bias + max_pool2d(3, stride 2, pad 1) (ConvSequence.conv, BinpackingNNet.py:34,39-40)
 * on channels-last x [B][H][W][Cin] -> out [B][(H+1)/2][(W+1)/2][32]; Cin 16 (<= 112 pixels) or 32 (<= 80 pixels). */
int rp_nn_convpool32(rp_ctx *ctx, const float *x_dev, const float *frag_dev, const float *bias_dev, float *out_dev, int64_t B, int32_t Cin, int32_t H,
                     int32_t W);
int rp_nn_bias_relu(rp_ctx *ctx, float *x_dev, const float *bias_dev, int64_t B, int32_t C, int32_t HW);
int rp_nn_bias_residual(rp_ctx *ctx, const float *x_dev, const float *bias_dev, const float *res_dev, float *out_dev, float *out_relu_dev,
                        int64_t B, int32_t C, int32_t HW);
int rp_nn_bias_pool(rp_ctx *ctx, const float *x_dev, const float *bias_dev, float *out_dev, float *out_relu_dev, int64_t B, int32_t C, int32_t H,
                    int32_t W, int32_t channels_last);
/* Host copy of the waiting leaves' packed states and slots (parity tests, host evaluators). */
int rp_leaf_states(rp_ctx *ctx, int32_t max_rows, uint64_t *rows_out /*[n][H]*/, uint8_t *remaining_out /*[n][N]*/,
                   int32_t *slot_out /*[n]*/, int32_t *n_out);
/* Expansion + backup: masks and renormalises pi with the leaf's valid moves (MCTS_bpp.py:88-100,
 * float64, NumPy summation order), stores Vs/Ns (:102-103) and backs v up the path (:130-139).
 * pi_dev [n][W*N] and v_dev [n] are DEVICE float32 (probabilities, i.e. exp(log_softmax), NNet.py:85),
 * row b belonging to the b-th waiting leaf.  pi must be what NNet.predict returns -- probabilities, pi >= 0: for a node with more
 * than 64 legal moves the search keeps "the unvisited move with the largest pi (lowest action among equals)" as the one PUCT
 * candidate of all unvisited moves, which is the reference's argmax over them exactly when cpuct > 0 and pi >= 0. */
int rp_commit_eval(rp_ctx *ctx, const float *pi_dev, const float *v_dev);
/* The same from the policy head's raw outputs (logits_fc, BinpackingNNet.py:69,79): the softmax of NNet.predict (NNet.py:81-85:
 * exp(log_softmax(x))) is taken inside the kernel, float32, exp(x - max) / sum -- no separate softmax pass over [n][W*N].
 * W * N <= 1536 (the row lives in LDS); larger action spaces take the softmax first and call rp_commit_eval. */
int rp_commit_eval_logits(rp_ctx *ctx, const float *logits_dev, const float *v_dev);
/* Same with HOST float32 buffers (tests, CPU evaluators). */
int rp_commit_eval_host(rp_ctx *ctx, const float *pi_host, const float *v_host, int32_t n_rows);

/* ---- results ------------------------------------------------------------------------- */
/* counts[a] = Nsa[(root, a)] (MCTS_bpp.py:40-41), host buffer [count][W*N] */
int rp_root_counts(rp_ctx *ctx, int32_t first, int32_t count, uint32_t *counts_out);
/* phase_out / sims_done_out / moves_out / episode_out: host buffers [count] (any may be NULL) */
int rp_game_status(rp_ctx *ctx, int32_t first, int32_t count, int32_t *phase_out, int32_t *sims_done_out,
                   int32_t *moves_out, uint64_t *episode_out);
/* Return value of each slot's latest simulation (MCTS.search returns v, MCTS_bpp.py:83,104,139) and its kind. */
int rp_last_values(rp_ctx *ctx, int32_t first, int32_t count, double *v_out, int32_t *kind_out);
/* Plays `action[i]` in slot first+i (CoachBPP.py:88-91): the child becomes the root, the tree is
 * kept; ended_out[i] = 0 / +-1 as getGameEnded, score_out[i] = r.  Only with RP_MOVE_EXTERNAL. */
int rp_advance_roots(rp_ctx *ctx, int32_t first, int32_t count, const int32_t *action, int32_t *ended_out,
                     double *score_out);
/* Finished episodes since the last call (auto move rules): up to max_n records, oldest first. */
int rp_pop_finished(rp_ctx *ctx, int64_t max_n, uint64_t *episode_id_out, int32_t *outcome_out, double *score_out,
                    int32_t *moves_out, int64_t *n_out);
/* Engine counters summed over slots since create/reset: [0] simulations, [1] expansions (leaf evaluations),
 * [2] terminal returns, [3] path edges walked, [4] sum of n_valid at selected nodes, [5] sum of n_valid at
 * expanded leaves, [6] transposition links, [7] nodes created, [8] moves played, [9] episodes finished,
 * [10] hash probes (64-slot windows), [11] key bytes compared, [12] sum of visited edges at selected nodes,
 * [13] visited-edge entries created, [14..15] reserved. */
int rp_counters(rp_ctx *ctx, int64_t *out16, int32_t reset);

/* ---- replay buffer (CoachBPP.executeEpisode's trainExamples, CoachBPP.py:80,99) -------- */
/* Number of examples recorded so far. */
int rp_examples_count(rp_ctx *ctx, int64_t *n_out);
/* Dense training tensors for examples [first, first+count): planes FP32 [count][N+1][H][W],
 * pi FP32 [count][W*N] (= counts / sum, MCTS_bpp.py:51-54), value FP32 [count] (the episode's ranked
 * outcome, 0 while the episode is unfinished) -- DEVICE buffers. */
int rp_examples_tensors(rp_ctx *ctx, int64_t first, int64_t count, float *planes_dev, float *pi_dev, float *value_dev);
/* Episode id and move number (0-based) of examples [first, first+count) -- HOST buffers.  The buffer fills in completion
 * order across slots; the reference appends episode by episode, move by move (CoachBPP.py:80,133), so callers sort by
 * (episode, move) before they trim to maxlenOfQueue (CoachBPP.py:122). */
int rp_examples_meta(rp_ctx *ctx, int64_t first, int64_t count, uint64_t *episode_id_out, int32_t *move_out);
int rp_examples_clear(rp_ctx *ctx);
/* The replay buffer in its PACKED form -- what ranks exchange and what the history of CoachBPP.learn keeps (CoachBPP.py:152-157), ~0.4 KB
 * per example at 20x20 / 32 items where the dense training tensors of rp_examples_tensors take 55 KB:
 *   key      u32 [E][KW]    the recorded state as the engine keys it: H row masks (u32 for W <= 32, else u64) then ceil(N / 32) words of
 *                           remaining-item bits (KW = H * (W > 32 ? 2 : 1) + ceil(N / 32), rounded up to even when W > 32)
 *   item_wh  u8  [E][N][2]  the episode's item sizes (getInitItems, BinPackingGame.py:37-51)
 *   value    i32 [E]        the episode's ranked outcome (CoachBPP.py:99)
 *   sp_off / sp_n i32 [E]   first entry and number of entries of the example's visit counts in the pool
 *   sp_act u16 / sp_cnt u32 [S]   pool of (action, Nsa) pairs, one per visited root edge (counts of MCTS_bpp.py:40-41; a one-hot (a, 1) for
 *                           greedy examples, :43-49)
 *   episode i64 / move i32 [E]    may be NULL
 * rp_examples_packed_count returns E and S; rp_examples_packed copies everything recorded so far into caller-owned DEVICE buffers of
 * those sizes (device-to-device, on the context's stream). */
int rp_examples_packed_count(rp_ctx *ctx, int64_t *n_examples_out, int64_t *n_sparse_out);
int rp_examples_packed(rp_ctx *ctx, int64_t n_examples, int64_t n_sparse, uint32_t *key_dev, uint8_t *item_wh_dev, int32_t *value_dev,
                       int32_t *sp_off_dev, int32_t *sp_n_dev, uint16_t *sp_act_dev, uint32_t *sp_cnt_dev, int64_t *episode_dev, int32_t *move_dev);
/* Training minibatch out of a packed replay set (the examples argument of NNetWrapper.train, NNet.py:27-67, picked by the index draw of
 * :43): row k of the outputs = example index_dev[k] (NULL: example k) of the caller-owned packed arrays above (sp_off 64-bit here: sets
 * gathered from several ranks and iterations outgrow 32 bits), expanded exactly like rp_examples_tensors.  Stateless: uses only the
 * context's geometry and stream. */
int rp_expand_examples(rp_ctx *ctx, int64_t n, const int64_t *index_dev, int64_t n_examples /* E */, int64_t n_sparse /* S */, const uint32_t *key_dev,
                       const uint8_t *item_wh_dev, const int32_t *value_dev, const int64_t *sp_off_dev, const int32_t *sp_n_dev,
                       const uint16_t *sp_act_dev, const uint32_t *sp_cnt_dev, float *planes_dev, float *pi_dev, float *value_out_dev);
/* Every index is checked on the device against E and S (an index list or a pool entry outside the arrays yields a zero row, never a
 * stray access); rp_check synchronises the context's stream and returns the first error device code recorded since the last check
 * (RP_ERR_ARG for such an index, RP_ERR_CAPACITY for an arena overflow), like every synchronising call does. */
int rp_check(rp_ctx *ctx);

/* ---- inspection (parity tests) --------------------------------------------------------- */
/* Sizes of slot g's tree: nodes in use and the span of its legal-move arena (the index range of rp_dump_tree's edge arrays). */
int rp_tree_size(rp_ctx *ctx, int32_t slot, int32_t *n_nodes_out, int32_t *n_edges_out);
/* High-water marks of the level arenas over all slots since create: chunks in use (legal-move runs, visited blocks) and
 * the chunk sizes in entries (6 and 32 bytes per entry).  Sizing aid for edge_cap / vis_cap. */
int rp_arena_peak(rp_ctx *ctx, int32_t *prior_chunks_out, int32_t *visited_chunks_out, int32_t *chunk_entries_out2);
/* Host copy of slot g's tree.  Node i: rows u64[H], remaining u8[N], term i8 (0 / +-1 = Es),
 * term_kind u8, expanded u8, ns u32, edge_off u32, n_valid u32.  Edge e (one per legal move, dense view of the sparse
 * device layout): action u16, P f64, Q f64, nsa u32, q_kind u8, child u32 (0xFFFFFFFF = not linked yet). */
int rp_dump_tree(rp_ctx *ctx, int32_t slot, uint64_t *node_rows, uint8_t *node_remaining, int8_t *node_term,
                 uint8_t *node_term_kind, uint8_t *node_expanded, uint32_t *node_ns, uint32_t *node_edge_off,
                 uint32_t *node_n_valid, uint16_t *edge_action, double *edge_p, double *edge_q, uint32_t *edge_nsa,
                 uint8_t *edge_q_kind, uint32_t *edge_child);
/* Device self-tests used by the GPU parity suite: sqrt / Q-update / NumPy-order sum on device,
 * host buffers in and out. */
int rp_selftest_sqrt(rp_ctx *ctx, int64_t n, double *sqrt_n_out, double *sqrt_n_eps_out);
int rp_selftest_q_update(rp_ctx *ctx, int64_t n, const double *q, const uint8_t *q_kind, const uint32_t *nsa,
                         const double *v, const uint8_t *v_kind, double *q_out, uint8_t *q_kind_out);
int rp_selftest_masked_prior(rp_ctx *ctx, int64_t B, const float *pi /*[B][A]*/, const uint8_t *valid /*[B][A]*/,
                             double *p_out /*[B][A]*/);

#ifdef __cplusplus
}
#endif
#endif
