// Gradient exchange of the data-parallel step (comm.hip): one object that owns the comm stream, its events and the backend --
// RCCL (loaded at run time, called from C) or host callbacks (tests: gloo) -- plus the engine-agnostic "what happens between two
// segments of a data-parallel step" logic both engines share.  NEW relative to the reference, which is single-device
// (/root/reference/global_configs.py:4,7; the DistributedSampler import at multimodal_driver.py:21 is never used).
#pragma once
#include <vector>
#include <utility>
#include "kernels.h"
#include "../../include/magbert_hip.h"

struct mb_comm {
    int rank = 0, world = 1;
    // backend: RCCL communicator (opaque ncclComm_t) or host callbacks
    void* nccl = nullptr;
    mb_all_reduce_cb ar_cb = nullptr;
    mb_all_gather_cb ag_cb = nullptr;
    void* ctx = nullptr;
    hipStream_t cs = nullptr;                 // the comm stream (created here: non-blocking; normal priority, MB_DP_COMM_PRIORITY=1: highest)
    std::vector<hipEvent_t> fork_ev;          // "the compute stream got this far": one per piece of a step, re-recorded every step
    hipEvent_t ev_layers = nullptr, ev_tail = nullptr;     // recorded on cs: every layer piece / every piece has been enqueued before
    hipEvent_t tev[4] = {nullptr, nullptr, nullptr, nullptr};   // timing events around the two places the compute stream waits for cs
    bool timing = false, tev_used[2] = {false, false};
    // scratch (caller-owned device memory, mb_comm_bind_scratch)
    int wire = mb::DT_F32;                    // wire format of the all-reduce pieces: fp32 (exact) or bf16 (staged through `stage`)
    char* scratch = nullptr; size_t scratch_bytes = 0;
    size_t n_params = 0; int vocab = 0, H = 0, cap = 0;
    size_t off_stage = 0, off_slot = 0, off_ids = 0, off_rows = 0;
    bool rows_ready = false;
    size_t pieces = 0, bytes_reduced = 0;     // statistics of the last step (tests / bench)
    // How the compute stream hands a finished segment to the comm stream (MB_DP_EVENT_MODE; same-box A/B, one-rank RCCL group,
    // profiles/r04_dp_event_modes.txt: single call 3.66 ms | mode 0: 3.81-3.83 | 1: 3.81-3.84 | 2: 3.64-3.70 | 3: 3.64-3.70):
    //   0 = hipEventRecord on the compute stream between two graph launches: each such marker costs the stream ~25 us;
    //   1 = the same with device-scope release events (no difference);
    //   2 = (default) the event is recorded by the LAST NODE of the segment's graph (dp_segment_end runs inside the capture): the
    //       graphs stay linear, nothing sits between two launches but the next launch;
    //   3 = ... and the optimizer segments start with wait nodes instead of stream waits (dp_segment_begin).
    // These nodes are ADDED TO THE CAPTURED GRAPH (hipGraphAddEventRecordNode / hipGraphAddEventWaitNode, dp_finish_segment_graph): a
    // plain hipEventRecord in a capturing stream only marks a dependency INSIDE the capture, adds no node and records nothing on
    // replay (tools/event_capture_probe.cpp: dependency held in 0 of 20 replays plain, 20 of 20 with a node;
    // profiles/r05_event_capture_probe.txt), and the in-capture external form (hipEventRecordWithFlags) is refused by the HIP 7.0
    // runtime a PyTorch process maps.  Round 4 shipped the plain form: its numbers above for modes 2 / 3 were measured WITHOUT the
    // dependency (re-measured in profiles/r05_dp_event_modes.txt).
    int event_mode = 2;
    // Gradient accumulation (multimodal_driver.py:375-376, 383-386): the step that follows micro-steps exchanges the word-embedding
    // table densely -- the rows its gradient holds are the union over the micro-steps, not this step's ids (mb_comm_set_row_exchange)
    bool rowwise = true;
    // Sharded optimizer update (mb_comm_set_sharding; NEW relative to the reference, which runs one AdamW over everything:
    // multimodal_driver.py:345, 384-386).  Every piece of layer GEMM weights the backward hands over (DpSpec::chunk) is cut into
    // `world` equal slices: the piece is REDUCE-SCATTERED instead of all-reduced (half the wire bytes; every rank receives its slice
    // of every piece, so all links carry every piece), each rank's AdamW covers its slices only (p, m, v of the other slices are not
    // touched: 28 B/parameter of HBM traffic less for (world-1)/world of 77 % of the model), and the operands of the next forward --
    // the bf16 shadow in bf16 mode, the fp32 parameters in parity mode -- come back by in-place all-gathers on the comm stream,
    // issued behind the optimizer launch that produced them; the next step's first launch waits for the last of them (ev_gather).
    // A remainder that does not divide by world x 256 elements stays replicated (all-reduced, updated by everyone).
    bool shard = false, gather_pending = false;
    hipEvent_t ev_gather = nullptr;
    // Sharded update with MORE THAN ONE piece ("cut" mode): the piece the backward finishes LAST -- the lowest layers -- is the one the next
    // forward needs FIRST, so it stays replicated (all-reduced, updated by every rank: no gather to wait for), and the forward of a
    // data-parallel step is cut at the same seams as its backward: `nf` forward-only segments in front of the backward segments, the
    // forward of piece k waits (host-side stream wait in front of its graph launch) for the all-gather of ITS piece only
    // (ev_chunk[k], recorded on the comm stream behind that gather).  The gathers run under the second optimizer segment and the
    // next step's first forward piece.  nf is set by the engines per step (0: forward in one piece inside the first segment).
    int nf = 0;
    std::vector<hipEvent_t> ev_chunk;
    std::vector<std::pair<size_t, size_t>> shard_chunks;     // the chunks of the last sharded step (mb_comm_gather_shards)
    size_t bytes_gathered = 0;
};

namespace mb {

// What a data-parallel step exchanges, in the order the backward finishes it.  The backward runs as `chunk.size()` segments; the
// gradient range chunk[s] is final when segment s has been enqueued.  All but the last go out "early" (the optimizer of those
// ranges starts as soon as THEY have landed); the last segment also ends the backward, so its range travels together with the
// tail -- everything that is not a layer's GEMM weight, of which the word-embedding table moves row-wise -- under the early
// ranges' optimizer launch.
struct DpSpec {
    std::vector<std::pair<size_t, size_t>> chunk;   // [begin, end) floats of the flat gradient buffer
    size_t tail_begin = 0, tail_end = 0;            // the rest: [tail_begin, tail_end)
    size_t word_off = 0;                            // the [vocab][H] word-embedding gradient inside the tail (rows = 0: dense)
    int word_rows = 0, H = 0;
    const int64_t* ids = nullptr; int T = 0;        // token ids of this rank's batch (device): the rows it touched
    char* gather_base = nullptr; int gather_es = 0; // sharded update: what the next forward reads of a chunk (bf16 shadow: 2 | fp32 parameters: 4)
    int n_sharded = 0;                              // sharded update: chunks [0, n_sharded) are sliced over the ranks, the rest stays replicated
};
// sharded update: how many of `nb` chunks are sliced -- all of a single piece; all but the last (lowest layers) of several
inline int dp_sharded_chunks(const mb_comm* c, int nb) { return !c->shard ? 0 : (nb > 1 ? nb - 1 : nb); }
// ... and how many forward-only segments a step of `nb` backward pieces then has in front
inline int dp_forward_segments(const mb_comm* c, int nb) { return (c->shard && nb > 1) ? nb - 1 : 0; }

// a chunk [b, e) of the flat buffers under the sharded update: this rank's slice, and the replicated remainder at the chunk's end
struct ShardSlice { size_t per, mine_b, mine_e, rem_b, rem_e; };
inline ShardSlice dp_shard_slice(const mb_comm* c, size_t b, size_t e) {
    ShardSlice s;
    s.per = (e - b) / (size_t)c->world / 256 * 256;          // slice boundaries on 1-KB marks (AdamW works on 16-byte quads)
    s.mine_b = b + (size_t)c->rank * s.per; s.mine_e = s.mine_b + s.per;
    s.rem_b = b + (size_t)c->world * s.per; s.rem_e = e;
    return s;
}

// layers per backward segment of a data-parallel step: MB_DP_CHUNKS="4,4,2,2" (must add up to n_layer) | MB_DP_CHUNK=n (uniform) |
// default: pieces of 4 layers (113 MB: few seams between the step's graphs) while the backward has plenty left to hide them, pieces of 2
// at the end, where what is still on the wire when the backward ends has to fit under the first optimizer launch
std::vector<int> dp_chunk_plan(int n_layer);

// segment numbering of a data-parallel step: [0, nb) the backward segments (nb = chunk.size(); the last one ends with the MAG /
// embedding stage) | nb: the optimizer over chunk[0 .. nb-2] | nb + 1: the optimizer over chunk[nb-1] and the tail.  Called by
// train_step_impl's `between` hook right after segment `seg` was enqueued on `st`.
int dp_between(mb_comm* c, const DpSpec& sp, float* G, int seg, hipStream_t st);
// called by the engines at the head / at the end of every segment's kernel sequence (inside the capture when the step is captured)
int dp_segment_begin(mb_comm* c, int nb, int seg, hipStream_t st);
int dp_segment_end(mb_comm* c, int nb, int seg, hipStream_t st);
// train_step_impl's post-capture hook (tag = the mb_comm, nseg = nb + 2): in event modes 2 / 3 appends the event-record node to a
// backward segment's graph, in mode 3 puts the wait-event node in front of an optimizer segment's graph, then checks the result
// (MB_ERR_MODE if the hand-off nodes are not where they must be)
int dp_finish_segment_graph(const void* tag, int nseg, int seg, hipGraph_t graph);

// at the head of a data-parallel step: the stream waits for the previous step's all-gathers (sharded update)
int dp_step_begin(mb_comm* c, hipStream_t st, bool cut);

// row-wise sum of a [vocab][H] fp32 table over the ranks (comm.hip): every rank touched the rows ids[0..T)
int comm_exchange_rows(mb_comm* c, float* table, const int64_t* ids, int T, hipStream_t s);
int comm_all_reduce(mb_comm* c, float* g, size_t count, hipStream_t s);

}  // namespace mb
