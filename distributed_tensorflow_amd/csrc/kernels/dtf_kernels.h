// C ABI of libdtf_kernels.so (gfx950 only).  Mirrored field-for-field by the
// ctypes structures in distributed_tensorflow_amd/ops/_lib.py.
#pragma once
#include "dtf_common.h"

// ---------------------------------------------------------------------------
// Device-resident step bookkeeping.  The training step never returns to the
// host: the last kernel of a step bumps the counters, appends one log record
// and selects the rows of the next mini-batch, so that n steps can be replayed
// from one hipGraph.  (Replaces the per-step global_step fetch, loss fetch and
// feed upload of /root/reference/distributed.py:137-149.)
// ---------------------------------------------------------------------------
struct DtfTailArgs {
  int* state;               // [0] = global_step (starts at 1, distributed.py:65), [1] = local_step
  int* log;                 // log_cap records of 4 x int32: {loss bits, correct, post_correct(prev batch), global_step fetched}
  const float* loss_part;   // per-workgroup partial loss sums written by the head kernel
  const int* correct_part;  // per-workgroup correct counts on the training rows
  const int* post_part;     // per-workgroup correct counts on the accuracy-only rows
  const int* ridx_cur;      // [2*B] dataset rows of the step that is finishing ([0,B) = its training batch)
  int* ridx_next;           // [2*B] written for the next step: [0,B) = new batch, [B,2B) = the batch just trained on
                            // (accuracy-only rows).  Must not alias ridx_cur: other workgroups of the same
                            // launch are still reading it.
  const int* perm;          // [2][n_train] double-buffered epoch permutations (buffer = epoch & 1)
  int nparts;
  int log_cap;
  int B;
  int n_train;
  int gs_inc;               // global_step increment per step
  int enable;               // 0 = no tail work at all
  float inv_b;              // 1 / B, turns the loss sum into the mean
  int pad_;
};

struct DtfSmallGemmArgs {
  const float* A;           // [M, K] row-major, leading dim lda
  const int* a_rows;        // optional gather: logical row i reads A row a_rows[i]
  const float* B;           // NN: [K, N] (ldb); NT: [N, K] (ldb)
  const float* bias;        // optional [N]
  const float* mask;        // optional [M, N]: output kept where mask > 0 (ReLU backward)
  float* C;                 // [M, N] (ldc)
  int M, N, K;
  int lda, ldb, ldc, ldmask;
  int relu;
  int b_transposed;
  unsigned a_bytes;         // bytes addressable from A / B (operands are read through 32-bit-offset buffer
  unsigned b_bytes;         // descriptors: each must stay below 4 GiB)
};

#define DTF_MAX_WGRAD_PROBLEMS 8
struct DtfWgradProblem {
  const float* X;           // [Kb, Mo] activations (rows optionally gathered through x_rows)
  const int* x_rows;
  const float* dY;          // [Kb, N]
  float* dW;                // [Mo, N] gradient out (mode 0) / unused (mode 1)
  float* db;                // [N]
  float* W;                 // [Mo, N] parameter updated in place (mode 1)
  float* b;                 // [N]
  int Mo, N, Kb;
  int ldx, ldy, ldw;
  int tile_start;           // first linear tile id of this problem
  int tiles_n;
  unsigned x_bytes;         // bytes addressable from X / dY
  unsigned y_bytes;
};

struct DtfWgradArgs {
  DtfWgradProblem prob[DTF_MAX_WGRAD_PROBLEMS];
  int nprob;
  int total_tiles;
  int mode;                 // 0: store gradients, 1: theta -= lr * grad fused in the epilogue
  float lr;
  DtfTailArgs tail;
};

struct DtfHeadArgs {
  const float* Hact;        // [M_total, Hd] last hidden activation (post-ReLU)
  const float* W2;          // [Hd, C]
  const float* b2;          // [C]
  const float* labels;      // [rows, C] one-hot / soft labels
  const int* rows;          // optional gather of label rows
  float* dz2;               // [M_train, 16] gradient wrt logits, zero padded to 16 columns
  float* dz1;               // [M_train, Hd] gradient wrt the hidden pre-activation (ReLU mask applied)
  float* probs;             // optional [M_total, C] softmax output y
  float* loss_part;
  int* correct_part;
  int* post_part;
  int M_total, M_train, Hd, C;
  int ldh, ld_dz1;
  int loss_mode;            // 0: reference (softmax applied twice, distributed.py:81+86), 1: standard xent on logits
  float inv_b;
  unsigned h_bytes;         // bytes addressable from Hact
  int pad_;
  void* dz1_bf;             // optional: dz1 as bf16 [M_train, Hd] (ld_bf) ...
  void* dz1_bf_t;           // ... and transposed [Hd, >= M_train] (ld_bf_t, a multiple of 4): the operands of the next
  int ld_bf, ld_bf_t;       //     bf16 products, written here instead of by a cast pass over the fp32 dz1 (which may be null)
};

// Output layer of a wide bf16 network (head_wide.hip): same mathematics as DtfHeadArgs with every row a training row,
// the last hidden activation given as the bf16 tensors the forward GEMM wrote.
struct DtfHeadWideArgs {
  const void* H;            // bf16 [M, ldh] last hidden activation (post-ReLU), row-major; ldh a multiple of 8
  const void* Ht;           // bf16 [Hd, ldht] the same transposed; ldht a multiple of 4
  const float* W2;          // [Hd, C]
  const float* b2;          // [C]
  const float* labels;      // [rows, C] one-hot / soft labels
  const int* rows;          // optional gather of label rows
  float* dz2;               // [M, 16] gradient wrt logits, zero padded to 16 columns
  void* dz1_bf;             // optional bf16 [M, ld_bf]: gradient wrt the hidden pre-activation (ReLU mask applied) ...
  void* dz1_bf_t;           // ... and transposed [Hd, ld_bf_t]
  float* loss_part;         // [ceil(M/16)]
  int* correct_part;
  int* post_part;           // optional, zeroed (no accuracy-only rows here)
  int M, Hd, C;             // M a multiple of 4, Hd a multiple of 8, C <= 16
  int ldh, ldht, ld_bf, ld_bf_t;
  int loss_mode;            // 0: reference (softmax applied twice), 1: standard cross-entropy on the logits
  float inv_b;
  int pad_;
};
DTF_EXPORT int dtf_head_wide(const DtfHeadWideArgs* a, hipStream_t s);

// dW2 = H^T dz2, db2 = colsum(dz2) for that layer, plus the step tail in one extra workgroup.
struct DtfOutWgradWideArgs {
  const void* Ht;           // bf16 [Hd, ldht], ldht a multiple of 8
  const float* dz2;         // [B, 16]
  float* dW;                // [Hd, C] (ldw)
  float* db;                // [C]
  int Hd, C, B;             // B a multiple of 8
  int ldht, ldw, pad_;
  DtfTailArgs tail;
};
DTF_EXPORT int dtf_out_wgrad_wide(const DtfOutWgradWideArgs* a, hipStream_t s);

// Fused softmax cross-entropy over [B, C] logits for any C (softmax_xent.hip): loss, gradient, accuracy count.
struct DtfXentArgs {
  const float* logits;        // [B, ld_logits]
  const float* labels_f;      // one-hot / soft labels [B, ld_labels], or null
  const long long* labels_i;  // class indices [B], or null (exactly one of the two label forms)
  float* dlogits;             // optional gradient [B, ld_dlogits] fp32
  void* dlogits_bf;           // optional gradient [B, ld_dlogits_bf] bf16 (operand of the backward products)
  float* loss_rows;           // optional per-row loss [B]
  float* loss_part;           // optional [ceil(B/4)] per-workgroup loss sums
  int* correct_part;          // optional [ceil(B/4)] per-workgroup counts of argmax(logits) == argmax(labels)
  int B, C;
  int ld_logits, ld_labels, ld_dlogits, ld_dlogits_bf;
  int loss_mode;              // 0: reference (softmax applied twice), 1: standard cross-entropy on the logits
  float grad_scale;           // usually 1 / B (mean loss)
  const int* label_rows;      // optional [B]: batch row i takes row label_rows[i] of the label arrays
};
DTF_EXPORT int dtf_softmax_xent(const DtfXentArgs* a, hipStream_t s);
DTF_EXPORT int dtf_softmax_xent_blocks(int B);

// Two-launch training step of the one-hidden-layer network (mlp2_fused.hip): launch 1 = hidden layer
// + per-column-tile partial logits, launch 2 = loss/accuracy records, softmax gradient, both weight
// gradients (optionally the SGD update), the step tail and the gather of the next batch.  One struct
// drives both launches.  Per-step buffers rotate over three slots; `q` is the slot of this step.
#define DTF_WLOG_INTS 64     // ints per record of the wide log: [0] global_step fetched, [1] local step,
                             // [4+rb] loss-sum bits, [20+rb] correct, [36+rb] post-update correct of row block rb
// Peer exchange inside launch 2 (mode 2; several ranks on one node, see p2p_allreduce.hip): every training
// workgroup stores its gradient tiles into this rank's exported slot, publishes "workgroup w of step s is
// complete" into every peer's flag array, waits for the same workgroup of every peer, sums the peers' tiles in
// rank order (remote reads over xGMI) and applies theta -= lr * sum - no separate all-reduce/update launch.
#define DTF_MLP2_MAX_WG 128   // training workgroups (dW1 + dW2 blocks) a flag array has room for
struct DtfMlp2Peer {
  const float* slot[8];     // this step's flat gradient slot in every rank's region (own rank: the local pointer)
  int* wgflags[8];          // arrival flags [3 slots][8 ranks][DTF_MLP2_MAX_WG] of every rank's region
  int* err;                 // set to 1 when a wait ran into its time limit
  int rank, world;
  unsigned timeout_ticks;   // of the 100 MHz constant clock
  int pad_;
};

// Sharded peer exchange inside launch 2 (mode 4): the parameter-server state is split by training workgroup,
// workgroup w (its weight-gradient tiles and the parameters under them) belonging to rank w % world.  Every rank
// computes every tile; a non-owner stores its tile into the owner's inbox and raises a flag there (reduce-scatter
// by direct stores over xGMI, one message per link), the owner adds the world contributions in rank order, applies
// the update and stores the new parameter tile into every peer's outbox (all-gather by direct stores), and the
// non-owners copy it into their replica.  Per rank and step: 2 x (world-1)/world of the gradient bytes leave over
// world-1 different links, instead of (world-1) x all of them arriving; two dependent hops instead of one.
// All regions have one layout, so the struct carries one base pointer per rank and the offsets.
struct DtfMlp2Shard {
  char* region[8];          // exported region of every rank (own rank: the local pointer)
  const float* params_base; // start of the local flat parameter buffer: offsets inside in/outboxes are parameter offsets
  int* err;                 // set to 1 when a wait ran into its time limit
  long long inbox_off;      // bytes from a region's start to float inbox[8 source ranks][n]
  long long outbox_off;     // ... to float outbox[n]
  long long inflag_off;     // ... to int inflags[8 source ranks][DTF_MLP2_MAX_WG]
  long long outflag_off;    // ... to int outflags[DTF_MLP2_MAX_WG]
  long long n;              // floats per flat buffer
  int rank, world;
  unsigned timeout_ticks;   // of the 100 MHz constant clock
  int adam;                 // 1: the owner applies Adam with the moments below (its ps-resident slot variables)
  // Adam on the owner: flat moment buffers of this rank (only the entries of tiles it owns are ever touched), the
  // true learning rate and the factor that turns the sum over the ranks into the aggregated gradient (1 / R);
  // update number = number of this step + adam_t_offset
  float* m;
  float* v;
  float adam_lr, grad_scale, b1, b2, eps;
  int adam_t_offset;
};

// Asynchronous push/pull inside launch 2 (mode 3; see the async section of p2p_allreduce.hip): every weight-
// gradient tile is applied to the owner's master slice through its mapped pointer (read at the start of the
// launch, written back minus lr * gradient: the lock-free read-modify-write of the reference's async mode) and
// kept as the local replica; the tail bumps the step counters with system-scope atomics.
struct DtfMlp2Async {
  float* master[8];         // master slice of every shard (own shard: local pointer)
  int* counters[8];         // per shard: [0] global_step (shard 0 only), [1] updates applied to this shard, [4 + rank] step clocks (shard 0)
  float* m[8];              // Adam: first / second moments next to every master slice (optimizer == 1)
  float* v[8];
  const float* params_base; // start of the local flat parameter buffer (W1, b1, W2, b2 are views into it)
  long long slice;          // floats per shard
  int shards;
  int optimizer;            // 0: SGD, 1: Adam with the owner-resident moments
  int pull_every;           // the replica takes the master's values on local steps divisible by this (1: every step);
                            // in between it takes this worker's own update only
  int bound;                // bounded staleness as in DtfAsyncPushPullArgs (< 0: none); needs astate
  int rank, world;
  unsigned timeout_ticks;   // how long a worker that is too far ahead waits for the slowest
  float b1, b2, eps;
  int* astate;              // the 16 ints of DtfAsyncPushPullArgs.astate; [6], [7]: "the step with this parity pulls whatever
                            // the period says", written one launch ahead (the launch that reads a slot never writes it)
  int* alog;                // optional [alog_cap][2] ring: {global_step after the push, other workers' updates the replica missed}
  int* applied_snap;        // Adam: [8] local ints, the slices' update counts as launch 1 of the step saw them
  int alog_cap;
  int pad_;
};

struct DtfMlp2Args {
  const float* X;           // [n, In] dataset rows (ldx)
  const float* labels;      // [n, C]
  float* Xg;                // [3][128][In] gathered batch rows (rows >= B stay zero); slot q = this step's batch, q-1 = the previous batch
  float* labg;              // [3][16][128] gathered label rows, class-major, zero padded
  int* ridx;                // [3][ridx_len] dataset row of each logical row ([0,B) batch, [B,2B) previous batch); host use
  float* H;                 // [h_rows >= max(R,128), ldh] hidden activations (written by launch 1)
  float* zpart;             // [16][R16*16][8] partial logits: (class, row, hidden column tile); zero-initialised
  float* W2s;               // [Hd*C] copy of W2 taken by launch 1 (launch 2 overwrites W2 while others read it)
  int* ctr_snap;            // [2] copy of state[0..1] taken by launch 1
  float* W1; float* b1; float* W2; float* b2;         // parameters; updated in place when mode == 1
  float* dW1; float* db1; float* dW2; float* db2;     // gradients out when mode == 0
  int* wlog;                // [log_cap][DTF_WLOG_INTS]
  int In, Hd, C, B, R;      // R = B, or 2B with the accuracy-only rows
  int ldx, ldh;
  int ridx_len;
  int h_rows;
  int q;                    // slot of this step, 0..2
  int loss_mode;
  int mode;                 // 0: store gradients, 1: theta -= lr * grad, 2: peer exchange (gradients -> slot, theta -= lr * sum over ranks),
                            // 3: async push/pull (master -= lr * grad through the mapped slices, replica = master),
                            // 4: sharded peer exchange (tile -> owner's inbox, owner updates and stores the tile into every outbox)
  int log_cap;
  float lr, inv_b;
  unsigned x_bytes;         // bytes addressable from X
  DtfTailArgs tail;         // state, perm, n_train, gs_inc (its log / partial / ridx fields are not used)
  DtfMlp2Peer peer;         // mode 2 only
  DtfMlp2Async async_;      // mode 3 only
  DtfMlp2Shard shard;       // mode 4 only
};
DTF_EXPORT int dtf_mlp2_fwd(const DtfMlp2Args* a, hipStream_t s);
DTF_EXPORT int dtf_mlp2_bwd(const DtfMlp2Args* a, hipStream_t s);
// Fills `out` for gradient slot `slot` of a connected peer-mapped communicator (dtf_p2p_create / dtf_p2p_connect).
DTF_EXPORT int dtf_p2p_mlp2_peer(void* comm, int slot, double timeout_s, DtfMlp2Peer* out);
// Fills `out` (except params_base) for the sharded exchange over the same communicator.
DTF_EXPORT int dtf_p2p_mlp2_shard(void* comm, double timeout_s, DtfMlp2Shard* out);

// One-launch pipelined step of the one-hidden-layer network for one worker with SGD (mlp_pipe.hip): the dependent
// path of a step works on Q = X' W1'(two steps old) and Gram matrices of the batches instead of the materialised
// weights; see the file header.  Buffers with two generations are indexed by the step's parity.
struct DtfPipeArgs {
  const float* X;           // [n, In] dataset rows (ldx)
  const float* labels;      // [n, C]
  float* Xa;                // [8][128][ldxa] gathered batch rows, column In = 1 (rows >= B and columns > In stay zero)
  float* labg;              // [8][16][128] gathered label rows, class-major, zero padded
  int* ridx;                // [8][ridx_len] dataset row of each logical row ([0,B) batch, [B,2B) the batch before it); host use
  float* Wm;                // [2][KAp * HP] W1 with b1 as row In, zero padded, in MFMA-fragment order per 16x16 tile
  float* W2m;               // [2][HP * 16] W2 with b2 as row Hd, zero padded, fragment order per 16 rows
  float* Q;                 // [2][16384] X'(s) W1'(s-2) - lr G(s,s-2) dz1(s-2), MFMA result layout per 16x16 tile
  float* Z1;                // [2][16384] z1 of a step before the ReLU (what the next step's accuracy-only rows start from), same layout
  int* labi;                // [8][128] class of each gathered row (argmax of its label row)
  float* GG;                // [2][3][16384] Gram matrices of column batch c: X'(c) X'(c)^T, X'(c+1) X'(c)^T, X'(c+2) X'(c)^T, A-operand layout
  float* D1;                // [2][16384] dz1 of a step (rows >= B zero), B-operand layout
  float* P2;                // [2][8][HP * 16] per row block: h'^T dz2, same order as W2m
  int* ctr;                 // [2][2] {global_step, local_step} a step reads (the tail writes the other generation)
  float* W1; float* b1; float* W2; float* b2;         // flat parameters: read by the prime launch, written by the flush launch
  int* wlog;                // [log_cap][DTF_WLOG_INTS]
  int In, Hd, C, B, R;      // R = B, or 2B with the accuracy-only rows
  int HP, KAp, ldxa;        // HP = Hd + 1 rounded up to 16, KAp = In + 1 rounded up to 16, ldxa = KAp
  int ldx, ridx_len;
  int q;                    // slot of the step, 0..7 (prime: slot before the first step; flush: slot after the last)
  int loss_mode, log_cap;
  int roles;                // bit set of workgroup roles in this launch
  int flush;                // 1: role W writes W1 / b1 of the flat buffer
  int prime;                // 1: role Q also builds Z1 of the batch before the first step
  int td;                   // tiles per side of the operand-layout matrices: 8 (Hd <= 127 and B <= 128) or 16 (Hd <= 256, B <= 256);
                            // with 16 every "128" above reads 256, every "16384" 65536 and P2 holds 16 row blocks
  float lr, inv_b;
  DtfTailArgs tail;         // state, perm, n_train, gs_inc
};
DTF_EXPORT int dtf_pipe_step(const DtfPipeArgs* a, hipStream_t s);
DTF_EXPORT int dtf_pipe_prime(const DtfPipeArgs* a, hipStream_t s);
DTF_EXPORT int dtf_pipe_flush(const DtfPipeArgs* a, hipStream_t s);
DTF_EXPORT int dtf_pipe_launch_roles(const DtfPipeArgs* a, hipStream_t s);

// One-launch pipelined step for N synchronous workers of one node (mlp_pipe_multi.hip): the workers exchange dz1 of
// their rows and their shares of the output layer's gradient through peer-mapped regions; see the file header.
//   b.Xa   [8][world][128][ldxa]   gathered batches of EVERY worker (every rank holds the data set and every shuffle)
//   b.ridx [8][world][ridx_len]    b.GG [2][4][world][16384]    b.D1, b.P2: unused (they live in the regions)
//   b.lr   = learning rate / world (the mean over the workers folded in)
// Region of a rank (uncached, IPC-exported; DTF_PIPEM_REGION_BYTES): D1 boxes [4][8][16384] floats | output-layer
// boxes [4][8][8][2048] floats | flags [8][8] ints (step stamps).
#define DTF_PIPEM_REGION_BYTES ((size_t)4 * 8 * 16384 * 4 + (size_t)4 * 8 * 8 * 2048 * 4 + 4096)
struct DtfPipeMultiArgs {
  DtfPipeArgs b;
  char* region[8];          // exchange region of every rank as mapped into this process (own rank: the allocation)
  const int* perm[8];       // [2][n_train] row order of every rank's shuffle (buffer = epoch & 1), on this device; null: none
  int* err;                 // device word: set to 1 when a wait for a peer ran into its limit
  int rank, world;
  unsigned timeout_ticks;   // of the 100 MHz constant clock
  int plain_order;          // 1: workgroup roles in plain block order (several ranks rehearsing on one GPU); 0: XCD-major
};
DTF_EXPORT int dtf_pipem_step(const DtfPipeMultiArgs* a, hipStream_t s);
DTF_EXPORT int dtf_pipem_prime(const DtfPipeMultiArgs* a, hipStream_t s);
DTF_EXPORT int dtf_pipem_flush(const DtfPipeMultiArgs* a, hipStream_t s);      // wait launch + flush launch
DTF_EXPORT int dtf_pipem_launch_roles(const DtfPipeMultiArgs* a, hipStream_t s);
DTF_EXPORT long long dtf_pipem_region_bytes(void);

// C[M,N] = A[M,K] . B[N,K]^T with both operands k-contiguous (see gemm_tiled.hip).
// dtype: 0 = fp32, 1 = bf16 (fp32 accumulate).  epilogue: 0 forward (+bias, ReLU, optional transposed copy Ct
// and fp32 copy C32), 1 backward (mask > 0, optional Ct), 2 weight gradient (C is always fp32; bias_out = row
// sums of B = column sums of dY).
struct DtfGemmArgs {
  const void* A;
  const void* B;
  void* C;
  void* Ct;
  float* C32;
  const float* bias;
  const void* mask;
  float* bias_out;
  int M, N, K;
  int lda, ldb, ldc, ldct, ldc32, ldmask;
  int relu;
  int epilogue;
  int dtype;
  unsigned a_bytes, b_bytes;
  const void* maskT;      // optional transposed mask [N, ldmaskt]: the 4 rows a lane owns are one 8/16-byte load
  int ldmaskt;
  int pad_;
  // Weight-gradient epilogue with the optimizer folded in (one worker, no aggregation; 8-wave kernel only): the
  // gradient tile never goes to memory, the epilogue reads master weights (and Adam moments) laid out like C,
  // writes them back updated and emits the compute-dtype copies W [M,N] (ld = ldc) and W^T [N, ldwt].
  float* opt_p;           // master weights [M, ldc]; null = plain weight gradient into C
  float* opt_m;           // Adam moments [M, ldc]; both null = SGD
  float* opt_v;
  void* opt_w;            // compute-dtype copy of the updated weights, optional
  void* opt_wt;           // ... transposed [N, ldwt], optional
  const int* opt_step;    // device counter of applied updates (Adam bias correction uses *opt_step + opt_step_offset)
  int opt_step_offset;
  int ldwt;
  float opt_lr, opt_b1, opt_b2, opt_eps;
  // Split-K weight gradient (8-wave kernel, plain EPI_WGRAD): a product whose 256x256 tiles would fill a fraction of
  // the chip (M = 784: 64 tiles) runs as ksplit x tiles workgroups, each over 1/ksplit of K; partial results go to
  // split_ws [ksplit][M][ldc] (and split_bias_ws [ksplit][N]) and a second launch sums them in split order into C
  // (bias_out), so the result does not depend on the order in which workgroups finish.
  int ksplit;             // 0 or 1 = off
  int pad2_;
  float* split_ws;
  float* split_bias_ws;   // needed when bias_out is given
};

// A pre-validated launch sequence: the hand-scheduled step (what a graph partitioner + executor do
// in the reference runtime).  One call enqueues every entry in order on `s`; stops at the first error
// and returns  (entry index << 8) | error code.
enum { DTF_OP_SMALL_GEMM = 1, DTF_OP_HEAD = 2, DTF_OP_WGRAD = 3, DTF_OP_TAIL = 4, DTF_OP_GEMM_TILED = 5, DTF_OP_UPDATE = 6,
       DTF_OP_MLP2_FWD = 7, DTF_OP_MLP2_BWD = 8, DTF_OP_P2P_UPDATE = 9, DTF_OP_ASYNC_PUSH_PULL = 10, DTF_OP_P2P_SYNC_GENERAL = 11, DTF_OP_SOFTMAX_XENT = 12 };

// Gradient aggregation + optimizer + parameter redistribution of one synchronous step.
enum { DTF_UPDATE_LOCAL = 0, DTF_UPDATE_ALLREDUCE = 1, DTF_UPDATE_SHARDED = 2 };
struct DtfUpdateArgs {
  void* comm;             // RCCL communicator (dtf_comm_init); null for DTF_UPDATE_LOCAL
  float* params;          // flat parameter replica [total]
  float* grads;           // flat gradients [total]
  float* rs_out;          // [shard_elems] reduce-scatter output (sharded mode)
  float* m;               // Adam state of the owned slice
  float* v;
  int* step;              // device counter of applied updates (Adam bias correction)
  int64_t total;
  int64_t shard_offset;
  int64_t shard_elems;
  int mode;
  int optimizer;          // 0 = SGD, 1 = Adam
  float lr, grad_scale, beta1, beta2, eps;
  int pad_;
};
DTF_EXPORT int dtf_run_update(const DtfUpdateArgs* a, hipStream_t s);
DTF_EXPORT int dtf_int_add(int* p, int v, hipStream_t s);
DTF_EXPORT int dtf_comm_load(const char* path);
DTF_EXPORT int dtf_comm_unique_id(char* out128);
DTF_EXPORT void* dtf_comm_init(const char* id128, int nranks, int rank, int* rc_out);
DTF_EXPORT int dtf_comm_destroy(void* comm);
DTF_EXPORT int dtf_comm_count(void* comm);
DTF_EXPORT const char* dtf_comm_error_string(int rc);
DTF_EXPORT int dtf_comm_all_reduce_f32(void* comm, float* buf, int64_t count, hipStream_t s);
DTF_EXPORT int dtf_comm_reduce_scatter_f32(void* comm, const float* send, float* recv, int64_t recvcount, hipStream_t s);
DTF_EXPORT int dtf_comm_all_gather_f32(void* comm, const float* send, float* recv, int64_t sendcount, hipStream_t s);
DTF_EXPORT int dtf_comm_broadcast_bytes(void* comm, void* buf, int64_t nbytes, int root, hipStream_t s);
// One-shot all-reduce + SGD over peer-mapped buffers (p2p_allreduce.hip).
struct DtfP2PUpdateArgs {
  void* comm;             // dtf_p2p_create / dtf_p2p_connect
  float* params;          // flat parameter replica, updated in place
  float* sum_out;         // optional: the summed gradient
  const int* seq_ptr;     // device counter with this step's number (identical on every rank, grows by one per step)
  double timeout_s;       // wall-clock bound of the arrival wait
  float scale;            // lr / R
  int slot;               // 0..2, the slot the gradients of this step were written to
};
// One gradient contribution of the general SyncReplicas path (replicas_to_aggregate != workers), see p2p_allreduce.hip.
struct DtfP2PGeneralArgs {
  void* comm;             // dtf_p2p_create / dtf_p2p_connect
  float* params;          // flat parameter replica, updated in place in the last round of an update
  float* acc;             // flat accumulator over the rounds of an update
  int* state;             // local counters: [0] global_step (+1 per update), [3] dropped gradients of this rank
  int* wlog;              // optional wide log: record of the round gets [2] accepted mask, [3] dropped so far
  const int* seq_ptr;     // device counter with the round number (identical on every rank, grows by one per round)
  double timeout_s;       // wall-clock bound of every wait
  double delay_s;         // test hook: this rank's contribution arrives this much later
  float scale;            // lr / R
  int slot;               // 0..2, the slot the gradients of this round were written to
  int need;               // contributions the accumulator takes in this round
  int first_round, last_round;
  int stamp_offset;       // test hook: stamp = round number + stamp_offset
  int log_cap;
  int pad_;
};
DTF_EXPORT int dtf_p2p_sync_general(const DtfP2PGeneralArgs* u, hipStream_t s);
DTF_EXPORT void* dtf_p2p_create(int rank, int world, int64_t n_floats, char* handle_out64);
DTF_EXPORT int dtf_p2p_connect(void* comm, const char* all_handles);
DTF_EXPORT float* dtf_p2p_slot_ptr(void* comm, int slot);
DTF_EXPORT int dtf_p2p_allreduce_sgd(void* comm, int slot, const int* seq_ptr, float* params, float* sum_out, float scale,
                                     double timeout_s, hipStream_t s);
DTF_EXPORT int dtf_p2p_update(const DtfP2PUpdateArgs* u, hipStream_t s);
DTF_EXPORT int dtf_p2p_error(void* comm);
DTF_EXPORT void dtf_p2p_destroy(void* comm);
// Asynchronous push/pull over peer-mapped parameter slices (p2p_allreduce.hip).
struct DtfAsyncPushPullArgs {
  float* master[8];       // slice s of the parameters in the (mapped) region of its owner
  int* counters[8];       // per owner: [0] global_step (meaningful for shard 0), [1] updates applied to the slice
  const float* grads;     // local flat gradients [shards * slice]
  float* params;          // local flat parameter replica, refreshed by the pull
  int* state;             // local counters: state[0] receives global_step
  int64_t slice;          // floats per slice (multiple of 4)
  int shards;
  int mode;               // 0: push (apply the optimizer to the masters) + pull, 1: pull only
  float lr;
  int optimizer;          // 0 = SGD, 1 = Adam (moments m[], v[] next to the master slices)
  float* m[8];            // Adam moments of slice s in the (mapped) region of its owner
  float* v[8];
  int* astate;            // optional local ints [16]: [0] global_step at the last pull, [1] own pushes since, [2] forced pulls,
                          // [3] pushes that had to wait for a slower worker, [4] "pull with the next push", [5] bound given up,
                          // [6], [7] the same decision per step parity for the push inside the backward launch (DtfMlp2Async),
                          // [8] workgroups of the running launch that have finished (the last one keeps the books)
  int* alog;              // optional ring [alog_cap][2]: {global_step after the push, updates of OTHER workers the replica missed}
  int alog_cap;
  int pull_every;         // pull on local steps divisible by this (<= 1: every step); other steps push only
  float b1, b2, eps;
  int bound;              // staleness bound (< 0: none): a replica that misses more than `bound` updates of other workers is
                          // refreshed with the next push, and a worker more than `bound` steps ahead of the slowest waits
  int rank, world;        // this worker and the number of workers (their step clocks sit at counters[0] + 4 + rank)
  unsigned timeout_ticks; // limit of that wait in ticks of the 100 MHz constant clock
};
DTF_EXPORT int dtf_async_push_pull(const DtfAsyncPushPullArgs* a, hipStream_t s);
DTF_EXPORT void* dtf_ipc_alloc(int64_t bytes, char* handle_out64);
DTF_EXPORT void* dtf_ipc_open(const char* handle64);
DTF_EXPORT int dtf_ipc_close(void* p);
DTF_EXPORT int dtf_ipc_free(void* p);
struct DtfScheduleEntry {
  int op;
  int flag;          // DTF_OP_TAIL: init_only
  const void* args;  // the op's argument struct
};
DTF_EXPORT int dtf_run_schedule(const DtfScheduleEntry* entries, int n, hipStream_t s);
DTF_EXPORT int dtf_run_schedule_steps(const DtfScheduleEntry* const* per_slot, const int* counts, int nslots, int first_slot,
                                      int n_steps, hipStream_t s);

DTF_EXPORT int dtf_gemm_tiled(const DtfGemmArgs* a, hipStream_t s);
DTF_EXPORT void dtf_gemm_set_modes(int big_tile, int pingpong);
// dstT[N, ldt] = src[M, lds]^T for 2-byte elements; M, N, lds, ldt multiples of 8, 16-byte aligned bases.
DTF_EXPORT int dtf_transpose16(const void* src, void* dstT, int M, int N, int lds_, int ldt, hipStream_t s);
DTF_EXPORT int dtf_gather_cast_transpose(const float* src, const int* rows, void* dst, void* dstT, int M, int N, int lds_,
                                         int ldd, int ldt, int dtype, hipStream_t s);
DTF_EXPORT int dtf_small_gemm(const DtfSmallGemmArgs* a, hipStream_t s);
DTF_EXPORT int dtf_wgrad_grouped(const DtfWgradArgs* a, hipStream_t s);
DTF_EXPORT int dtf_head(const DtfHeadArgs* a, hipStream_t s);
DTF_EXPORT int dtf_step_tail(const DtfTailArgs* a, int init_only, hipStream_t s);

// flat fused optimizers (shard updates)
DTF_EXPORT int dtf_sgd_apply(float* param, const float* grad, int64_t n, float lr, float grad_scale, hipStream_t s);
#define DTF_MAX_OPT_RANGES 16
struct DtfOptRangesArgs {
  float* p; const float* g; float* m; float* v;      // flat buffers; m, v null = SGD
  const int* step_ptr;                               // Adam: number of this update = *step_ptr + step_offset
  long long lo[DTF_MAX_OPT_RANGES], hi[DTF_MAX_OPT_RANGES];   // element ranges, multiples of 4
  int nranges, step_offset;
  float lr, b1, b2, eps, gscale;
  int pad_;
};
DTF_EXPORT int dtf_opt_ranges(const DtfOptRangesArgs* a, hipStream_t s);
DTF_EXPORT int dtf_adam_apply(float* param, const float* grad, float* m, float* v, int64_t n, float lr, float beta1,
                              float beta2, float eps, float grad_scale, const int* step_ptr, int step_offset,
                              hipStream_t s);
DTF_EXPORT int dtf_truncated_normal(float* out, int64_t n, float stddev, uint64_t seed, uint64_t stream_id, hipStream_t s);
DTF_EXPORT int dtf_accumulate(float* dst, const float* src, int64_t n, float alpha, float beta, hipStream_t s);
DTF_EXPORT int dtf_upload_words(void* dst, const void* host_src, int64_t n_words, hipStream_t s);
DTF_EXPORT int dtf_kernels_abi_version();
