// ovs_bow.hip -- descriptors to BowVector + FeatureVector on the device: frame::compute_bow /
// keyframe::compute_bow (src/openvslam/data/frame.cc:162-170, data/keyframe.cc:181-189), i.e.
// DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>::transform(features, v, fv, levelsup)
// (data/bow_vocabulary.h:16).  DBoW2 is a dependency of the reference that is not in its tree
// (shinsumicco/DBoW2@687fcb7, Dockerfile.desktop:144); the kernels follow its published algorithm, which
// include/ovs.h states next to the entry points.  The results feed match::bow_tree and
// robust::match_for_triangulation (match/bow_tree.cc:25-35, match/robust.cc:44-52), which walk the two
// FeatureVectors node by node.
//
// Three launches per batch of descriptor sets:
//   k_bow_descend   16 lanes per descriptor walk the tree: every lane takes one child of the current node
//                   (32 bytes, two 16-byte loads), the group minimum of `distance << 22 | child position`
//                   is the reference's "strictly smaller wins, first child on ties"; a chain of
//                   depth x (offsets -> child id -> descriptor) dependent loads per descriptor, so the
//                   kernel is latency bound and wants many descriptors in flight, which a batch gives it;
//   k_bow_rank      std::map order without a sort: the rank of a descriptor among the (node id, index) and
//                   (word id, index) keys of its frame by counting, keys tiled through LDS;
//   k_bow_compact   one workgroup per frame turns the two sorted sequences into the flattened maps
//                   (head flags + block scan), forms the BowVector values with DBoW2's additions in
//                   DBoW2's order and normalises them (the norm is a sequential sum in ascending word
//                   order, as iterating the std::map gives it: one lane adds, everybody divides).
#include <algorithm>
#include <cmath>
#include <memory>

#include "ovs_match_common.h"

using namespace ovsm;

struct ovs_bow_vocab {
    int device = 0;
    int depth_levels = 0, weighting = 0, scoring = 0, num_nodes = 0, max_depth = 0;
    bool empty = true;
    ovs::dev_buffer node_descs, child_offsets, children, word_ids, weights;
};

namespace {

// sort key of a descriptor whose word has weight <= 0: behind every node id (< 2^31) and every word id (the
// vocabulary is refused when a word id is 0xFFFFFFFF)
constexpr u32 kStopped = 0xFFFFFFFFu;
constexpr u32 kChildPosBits = 22;

struct bow_tree_dev {
    const uint8_t* node_descs;
    const u32* child_offsets;
    const u32* children;
    const u32* word_ids;
    const double* weights;
    int nid_level; // depth_levels - levelsup
    int max_depth;
};

// minimum over the 16 lanes of a DPP row, in every lane of the row
__device__ __forceinline__ u32 row_min_u32(u32 v) {
    v = min(v, (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xf, 0xf, false));  // quad_perm [1,0,3,2]
    v = min(v, (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xf, 0xf, false));  // quad_perm [2,3,0,1]
    v = min(v, (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xf, 0xf, false)); // row_half_mirror
    v = min(v, (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xf, 0xf, false)); // row_mirror
    return v;
}

__global__ __launch_bounds__(256) void k_bow_descend(bow_tree_dev t, const uint8_t* __restrict__ descs, const int* __restrict__ counts, int fixed_n,
                                                     int cap, u32* __restrict__ word_of, u32* __restrict__ node_of, double* __restrict__ weight_of,
                                                     u32* __restrict__ key_node, u32* __restrict__ key_word) {
    const int frame = blockIdx.y;
    const int n = min(counts ? counts[frame] : fixed_n, cap);
    const int i = blockIdx.x * 16 + (threadIdx.x >> 4);
    const u32 sub = threadIdx.x & 15;
    if (i >= n) {
        return; // whole rows leave together
    }
    const size_t o = (size_t)frame * cap + i;
    u32 q[8];
    load_desc(descs + o * 32, q);
    u32 node = 0, nid = 0;
    bool nid_set = t.nid_level <= 0; // the root
    int level = 0;
    // TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup): do { ... } while (!isLeaf())
    for (int step = 0; step < t.max_depth; ++step) {
        const u32 off = t.child_offsets[node], end = t.child_offsets[node + 1];
        if (off == end) {
            break;
        }
        ++level;
        u32 best = 0xFFFFFFFFu;
        for (u32 c0 = off; c0 < end; c0 += 16) {
            const u32 c = c0 + sub;
            if (c < end) {
                u32 d[8];
                load_desc(t.node_descs + (size_t)t.children[c] * 32, d);
                best = min(best, (hamming8(q, d) << kChildPosBits) | (c - off));
            }
        }
        best = row_min_u32(best);
        node = t.children[off + (best & ((1u << kChildPosBits) - 1u))];
        if (level == t.nid_level) {
            nid = node;
            nid_set = true;
        }
    }
    if (!nid_set) {
        nid = node;
    }
    if (sub == 0) {
        const double w = t.weights[node];
        const bool stopped = !(w > 0);
        const u32 word = t.word_ids[node];
        if (word_of) {
            word_of[o] = stopped ? 0xFFFFFFFFu : word;
        }
        if (node_of) {
            node_of[o] = nid;
        }
        weight_of[o] = w;
        key_node[o] = stopped ? kStopped : (u32)nid;
        key_word[o] = stopped ? kStopped : (u32)word;
    }
}

// rank of descriptor i among the keys of its frame in (key, index) order: the position std::map iteration
// followed by the per-key vectors (indices pushed in ascending order) gives it.  A workgroup ranks 64
// descriptors (one per lane); the keys of the frame pass through LDS in tiles of 256, and each of the four
// waves compares its lanes' keys with a quarter of every tile (all lanes of a wave read the same LDS word: a
// broadcast), so a frame of 2000 descriptors is 32 workgroups of 500 steps instead of 8 of 2000.  Only the
// tile that holds the workgroup's own descriptors needs the index comparison; in the tiles before it an
// equal key counts, in the tiles behind it it does not.
constexpr int kRankLanes = 64, kRankTile = 256;
__global__ __launch_bounds__(256) void k_bow_rank(const int* __restrict__ counts, int fixed_n, int cap, const u32* __restrict__ key_node,
                                                  const u32* __restrict__ key_word, const double* __restrict__ weight_of,
                                                  u32* __restrict__ sorted_node, u32* __restrict__ sorted_node_idx, u32* __restrict__ sorted_word,
                                                  double* __restrict__ sorted_word_weight) {
    __shared__ uint2 s_keys[kRankTile]; // x: node key, y: word key
    __shared__ u32 s_part[2][4][kRankLanes];
    const int frame = blockIdx.y;
    const int n = min(counts ? counts[frame] : fixed_n, cap);
    const int i0 = (int)blockIdx.x * kRankLanes;
    if (i0 >= n) {
        return;
    }
    const size_t fo = (size_t)frame * cap;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = i0 + lane;
    const u32 kn = i < n ? key_node[fo + i] : kStopped;
    const u32 kw = i < n ? key_word[fo + i] : kStopped;
    u32 rn = 0, rw = 0;
    for (int j0 = 0; j0 < n; j0 += kRankTile) {
        const int j = j0 + tid;
        __syncthreads();
        s_keys[tid] = make_uint2(j < n ? key_node[fo + j] : kStopped, j < n ? key_word[fo + j] : kStopped);
        __syncthreads();
        const int m = min(kRankTile, n - j0);
        const int kb = wave * (kRankTile / 4), ke = min(kb + kRankTile / 4, m);
        if (j0 + kRankTile <= i0) {
            // every key of the tile has a smaller index than the workgroup's descriptors
            for (int k = kb; k < ke; ++k) {
                const uint2 o = s_keys[k];
                rn += o.x <= kn;
                rw += o.y <= kw;
            }
        }
        else if (i0 + kRankLanes <= j0) {
            for (int k = kb; k < ke; ++k) {
                const uint2 o = s_keys[k];
                rn += o.x < kn;
                rw += o.y < kw;
            }
        }
        else {
            for (int k = kb; k < ke; ++k) {
                const uint2 o = s_keys[k];
                const bool before = j0 + k < i;
                rn += (o.x < kn) || (o.x == kn && before);
                rw += (o.y < kw) || (o.y == kw && before);
            }
        }
    }
    s_part[0][wave][lane] = rn;
    s_part[1][wave][lane] = rw;
    __syncthreads();
    if (wave == 0 && i < n) {
        rn = s_part[0][0][lane] + s_part[0][1][lane] + s_part[0][2][lane] + s_part[0][3][lane];
        rw = s_part[1][0][lane] + s_part[1][1][lane] + s_part[1][2][lane] + s_part[1][3][lane];
        sorted_node[fo + rn] = kn;
        sorted_node_idx[fo + rn] = (u32)i;
        sorted_word[fo + rw] = kw;
        sorted_word_weight[fo + rw] = weight_of[fo + i];
    }
}

constexpr int kCompactThreads = 1024;
constexpr int kNormChunk = 4096; // terms of the norm staged in LDS at a time (32 KB)

// inclusive scan over the workgroup; `red` holds one word per wave plus the total
__device__ __forceinline__ u32 block_incl_scan_u32(u32 x, u32* red, u32& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32 inc = wave_incl_scan_u32(x);
    __syncthreads(); // red may still be read by the previous call
    if (lane == 63) {
        red[wave] = inc;
    }
    __syncthreads();
    u32 before = 0, tot = 0;
    for (int w = 0; w < kCompactThreads / 64; ++w) {
        const u32 v = red[w];
        before += w < wave ? v : 0u;
        tot += v;
    }
    total = tot;
    return before + inc;
}

__global__ __launch_bounds__(kCompactThreads) void k_bow_compact(const int* __restrict__ counts, int fixed_n, int cap, int weighting, int scoring,
                                                                 const u32* __restrict__ key_node, const u32* __restrict__ sorted_node,
                                                                 const u32* __restrict__ sorted_node_idx, const u32* __restrict__ sorted_word,
                                                                 const double* __restrict__ sorted_word_weight, u32* __restrict__ word_start,
                                                                 u32* __restrict__ bow_word_ids, double* __restrict__ bow_values, int* __restrict__ num_words_out,
                                                                 u32* __restrict__ fv_node_ids, u32* __restrict__ fv_offsets, u32* __restrict__ fv_indices,
                                                                 int* __restrict__ fv_num_nodes_out) {
    __shared__ u32 red[kCompactThreads / 64];
    __shared__ double s_norm;
    __shared__ double s_terms[kNormChunk];
    const int frame = blockIdx.x, tid = threadIdx.x;
    const int n = min(counts ? counts[frame] : fixed_n, cap);
    const size_t fo = (size_t)frame * cap;
    key_node += fo;
    sorted_node += fo;
    sorted_node_idx += fo;
    sorted_word += fo;
    sorted_word_weight += fo;
    word_start += (size_t)frame * (cap + 1);
    bow_word_ids += fo;
    bow_values += fo;
    fv_node_ids += fo;
    fv_offsets += (size_t)frame * (cap + 1);
    fv_indices += fo;

    // descriptors that are not stopped: the first M entries of both sorted sequences
    u32 M = 0;
    {
        u32 mine = 0;
        for (int i = tid; i < n; i += kCompactThreads) {
            mine += key_node[i] != kStopped;
        }
        block_incl_scan_u32(mine, red, M);
    }
    // FeatureVector: one entry per distinct node id, indices in ascending order behind it
    u32 num_nodes = 0;
    for (u32 r0 = 0; r0 < M; r0 += kCompactThreads) {
        const u32 r = r0 + tid;
        const bool head = r < M && (r == 0 || sorted_node[r] != sorted_node[r - 1]);
        u32 total;
        const u32 incl = block_incl_scan_u32(head ? 1u : 0u, red, total);
        if (head) {
            const u32 slot = num_nodes + incl - 1;
            fv_node_ids[slot] = sorted_node[r];
            fv_offsets[slot] = r;
        }
        if (r < M) {
            fv_indices[r] = sorted_node_idx[r];
        }
        num_nodes += total;
    }
    // BowVector: one entry per distinct word id
    u32 num_words = 0;
    for (u32 r0 = 0; r0 < M; r0 += kCompactThreads) {
        const u32 r = r0 + tid;
        const bool head = r < M && (r == 0 || sorted_word[r] != sorted_word[r - 1]);
        u32 total;
        const u32 incl = block_incl_scan_u32(head ? 1u : 0u, red, total);
        if (head) {
            const u32 slot = num_words + incl - 1;
            bow_word_ids[slot] = sorted_word[r];
            word_start[slot] = r;
        }
        num_words += total;
    }
    if (tid == 0) {
        fv_offsets[num_nodes] = M;
        word_start[num_words] = M;
        fv_num_nodes_out[frame] = (int)num_nodes;
        num_words_out[frame] = (int)num_words;
    }
    __syncthreads();
    // L1Scoring, L2Scoring, ChiSquareScoring, KLScoring, BhattacharyyaScoring normalise; DotProductScoring does not
    const bool tf = weighting == OVS_BOW_TF || weighting == OVS_BOW_TF_IDF;
    const bool must = scoring != OVS_BOW_DOT_PRODUCT;
    const bool l2 = scoring == OVS_BOW_L2_NORM;
    for (u32 s = tid; s < num_words; s += kCompactThreads) {
        const u32 b = word_start[s], e = word_start[s + 1];
        const double w = sorted_word_weight[b];
        double v = w; // BowVector::addWeight inserts w, then adds w per further descriptor; addIfNotExist keeps w
        if (tf) {
            for (u32 c = b + 1; c < e; ++c) {
                v = __dadd_rn(v, w);
            }
            if (!must) {
                v = __ddiv_rn(v, (double)num_words);
            }
        }
        bow_values[s] = v;
    }
    __syncthreads();
    if (must) {
        // BowVector::normalize: the norm is accumulated in map order, a chain of dependent double additions that
        // one lane walks.  The terms come to it through LDS, a chunk at a time (everybody loads, one lane adds:
        // read from global memory one by one, the 2000 values of a frame cost 0.1 ms of load latency).
        double norm = 0.0;
        for (u32 c0 = 0; c0 < num_words; c0 += kNormChunk) {
            const u32 m = min((u32)kNormChunk, num_words - c0);
            __syncthreads(); // the previous chunk has been added
            for (u32 s = tid; s < m; s += kCompactThreads) {
                const double v = bow_values[c0 + s];
                s_terms[s] = l2 ? __dmul_rn(v, v) : fabs(v);
            }
            __syncthreads();
            if (tid == 0) {
                u32 s = 0;
                for (; s + 8 <= m; s += 8) {
                    double t[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        t[u] = s_terms[s + u];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        norm = __dadd_rn(norm, t[u]);
                    }
                }
                for (; s < m; ++s) {
                    norm = __dadd_rn(norm, s_terms[s]);
                }
            }
        }
        if (tid == 0) {
            s_norm = l2 ? __dsqrt_rn(norm) : norm;
        }
        __syncthreads();
        norm = s_norm;
        if (norm > 0.0) {
            for (u32 s = tid; s < num_words; s += kCompactThreads) {
                bow_values[s] = __ddiv_rn(bow_values[s], norm);
            }
        }
    }
}

int check_vocab(const ovs_bow_vocab* v) {
    if (!v) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "null vocabulary");
    }
    return OVS_OK;
}

// scratch per descriptor slot of a batch, carved from an arena
struct bow_scratch {
    double* weight_of;
    u32* key_node;
    u32* key_word;
    u32* sorted_node;
    u32* sorted_node_idx;
    u32* sorted_word;
    double* sorted_word_weight;
    u32* word_start;
    static size_t bytes(size_t frames, size_t cap) {
        const size_t fc = frames * cap;
        return al(fc * 8) + 5 * al(fc * 4) + al(fc * 8) + al(frames * (cap + 1) * 4);
    }
    void carve(carver& cv, size_t frames, size_t cap) {
        const size_t fc = frames * cap;
        weight_of = cv.take<double>(fc);
        key_node = cv.take<u32>(fc);
        key_word = cv.take<u32>(fc);
        sorted_node = cv.take<u32>(fc);
        sorted_node_idx = cv.take<u32>(fc);
        sorted_word = cv.take<u32>(fc);
        sorted_word_weight = cv.take<double>(fc);
        word_start = cv.take<u32>(frames * (cap + 1));
    }
};

int launch_transform(const ovs_bow_vocab* v, const uint8_t* d_descs, const int32_t* d_counts, int fixed_n, int n_frames, int cap, int levelsup,
                     u32* d_word_of, u32* d_node_of, const bow_scratch& s, u32* d_bow_word_ids, double* d_bow_values, int32_t* d_num_words,
                     u32* d_fv_node_ids, u32* d_fv_offsets, u32* d_fv_indices, int32_t* d_fv_num_nodes, hipStream_t st) {
    bow_tree_dev t;
    t.node_descs = v->node_descs.as<uint8_t>();
    t.child_offsets = v->child_offsets.as<u32>();
    t.children = v->children.as<u32>();
    t.word_ids = v->word_ids.as<u32>();
    t.weights = v->weights.as<double>();
    t.nid_level = v->depth_levels - levelsup;
    t.max_depth = v->max_depth;
    OVS_LAUNCH("k_bow_descend", k_bow_descend, dim3((cap + 15) / 16, n_frames), dim3(256), 0, st, t, d_descs, d_counts, fixed_n, cap, d_word_of, d_node_of,
               s.weight_of, s.key_node, s.key_word);
    OVS_LAUNCH("k_bow_rank", k_bow_rank, dim3((cap + kRankLanes - 1) / kRankLanes, n_frames), dim3(256), 0, st, d_counts, fixed_n, cap, s.key_node, s.key_word, s.weight_of,
               s.sorted_node, s.sorted_node_idx, s.sorted_word, s.sorted_word_weight);
    OVS_LAUNCH("k_bow_compact", k_bow_compact, dim3(n_frames), dim3(kCompactThreads), 0, st, d_counts, fixed_n, cap, v->weighting, v->scoring, s.key_node,
               s.sorted_node, s.sorted_node_idx, s.sorted_word, s.sorted_word_weight, s.word_start, d_bow_word_ids, d_bow_values, d_num_words,
               d_fv_node_ids, d_fv_offsets, d_fv_indices, d_fv_num_nodes);
    OVS_HIP(hipGetLastError());
    return OVS_OK;
}

} // namespace

extern "C" {

int ovs_bow_vocab_create(const ovs_bow_vocab_view* view, int device, ovs_bow_vocab** out) {
    if (!view || !out) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "null argument");
    }
    const long long nn = view->num_nodes;
    if (nn < 1 || !view->child_offsets || (nn > 1 && (!view->node_descs || !view->children || !view->word_ids || !view->weights))) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "vocabulary view: missing arrays");
    }
    if (view->weighting < OVS_BOW_TF_IDF || view->weighting > OVS_BOW_BINARY || view->scoring < OVS_BOW_L1_NORM || view->scoring > OVS_BOW_DOT_PRODUCT) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "vocabulary view: unknown weighting or scoring type");
    }
    // a tree rooted at node 0: num_nodes - 1 child entries, every other node named once, all reachable
    const uint32_t* co = view->child_offsets;
    if (co[0] != 0 || co[nn] != static_cast<uint32_t>(nn - 1)) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "vocabulary view: child_offsets do not describe num_nodes - 1 children");
    }
    for (long long i = 0; i < nn; ++i) {
        if (co[i + 1] < co[i] || co[i + 1] - co[i] >= (1u << kChildPosBits)) {
            return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "vocabulary view: child_offsets not ascending (or 2^22 children of one node)");
        }
    }
    for (long long i = 0; i < nn && nn > 1; ++i) {
        if (view->word_ids[i] == 0xFFFFFFFFu) {
            return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "vocabulary view: word id 0xFFFFFFFF is not supported");
        }
    }
    std::vector<uint8_t> seen(nn, 0);
    for (long long c = 0; c < nn - 1; ++c) {
        const uint32_t id = view->children[c];
        if (id == 0 || id >= nn || seen[id]) {
            return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "vocabulary view: a child id is the root, out of range or named twice");
        }
        seen[id] = 1;
    }
    int max_depth = 0;
    {
        std::vector<uint32_t> level{0}, next;
        long long reached = 1;
        while (true) {
            next.clear();
            for (uint32_t n : level) {
                next.insert(next.end(), view->children + co[n], view->children + co[n + 1]);
            }
            if (next.empty()) {
                break;
            }
            reached += static_cast<long long>(next.size());
            ++max_depth;
            level.swap(next);
        }
        if (reached != nn) {
            return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "vocabulary view: nodes that the root does not reach");
        }
    }
    int rc = check_device(device);
    if (rc != OVS_OK) {
        return rc;
    }
    ovs::device_guard guard(device);
    std::unique_ptr<ovs_bow_vocab> v(new ovs_bow_vocab());
    v->device = device;
    v->depth_levels = view->depth_levels;
    v->weighting = view->weighting;
    v->scoring = view->scoring;
    v->num_nodes = view->num_nodes;
    v->max_depth = max_depth;
    v->empty = nn == 1;
    auto up = [&](ovs::dev_buffer& b, const void* src, size_t bytes) -> int {
        int r = b.reserve(std::max<size_t>(bytes, 256));
        if (r != OVS_OK) {
            return r;
        }
        if (bytes) {
            OVS_HIP(hipMemcpy(b.ptr, src, bytes, hipMemcpyHostToDevice));
        }
        return OVS_OK;
    };
    const size_t n = static_cast<size_t>(nn);
    if ((rc = up(v->child_offsets, co, (n + 1) * 4)) != OVS_OK || (rc = up(v->children, view->children, (n - 1) * 4)) != OVS_OK
        || (rc = up(v->node_descs, view->node_descs, n > 1 ? n * 32 : 0)) != OVS_OK || (rc = up(v->word_ids, view->word_ids, n > 1 ? n * 4 : 0)) != OVS_OK
        || (rc = up(v->weights, view->weights, n > 1 ? n * 8 : 0)) != OVS_OK) {
        ovs_bow_vocab_destroy(v.release());
        return rc;
    }
    *out = v.release();
    return OVS_OK;
}

void ovs_bow_vocab_destroy(ovs_bow_vocab* v) {
    if (!v) {
        return;
    }
    ovs::device_guard guard(v->device);
    v->node_descs.release();
    v->child_offsets.release();
    v->children.release();
    v->word_ids.release();
    v->weights.release();
    delete v;
}

int ovs_bow_transform_batch_dev(const ovs_bow_vocab* v, const uint8_t* d_descs, const int32_t* d_counts, int n_frames, int cap, int levelsup,
                                uint32_t* d_word_of_feature, uint32_t* d_node_of_feature, uint32_t* d_bow_word_ids, double* d_bow_values,
                                int32_t* d_num_words, uint32_t* d_fv_node_ids, uint32_t* d_fv_offsets, uint32_t* d_fv_indices, int32_t* d_fv_num_nodes,
                                void* stream_) {
    int rc = check_vocab(v);
    if (rc != OVS_OK) {
        return rc;
    }
    if (n_frames <= 0 || n_frames > 65535 || cap <= 0 || !d_descs || !d_bow_word_ids || !d_bow_values || !d_num_words || !d_fv_node_ids || !d_fv_offsets
        || !d_fv_indices || !d_fv_num_nodes) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "invalid argument");
    }
    if (cap >= (1 << 20)) {
        return ovs::fail(OVS_ERR_UNSUPPORTED, "more than 2^20 descriptors per frame are not supported");
    }
    if (reinterpret_cast<uintptr_t>(d_descs) & 15) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "descriptors must be 16-byte aligned");
    }
    ovs::device_guard guard(v->device);
    hipStream_t st = static_cast<hipStream_t>(stream_);
    if (v->empty) {
        // TemplatedVocabulary::transform: v.clear(); fv.clear(); if (empty()) return;
        OVS_HIP(hipMemsetAsync(d_num_words, 0, static_cast<size_t>(n_frames) * 4, st));
        OVS_HIP(hipMemsetAsync(d_fv_num_nodes, 0, static_cast<size_t>(n_frames) * 4, st));
        return OVS_OK;
    }
    thread_ctx* ctx = ctx_for_stream(v->device, st);
    if ((rc = ctx->arena.reserve(bow_scratch::bytes(n_frames, cap))) != OVS_OK) {
        return rc;
    }
    carver cv;
    cv.base = ctx->arena.as<uint8_t>();
    bow_scratch s;
    s.carve(cv, n_frames, cap);
    return launch_transform(v, d_descs, d_counts, cap, n_frames, cap, levelsup, d_word_of_feature, d_node_of_feature, s, d_bow_word_ids, d_bow_values,
                            d_num_words, d_fv_node_ids, d_fv_offsets, d_fv_indices, d_fv_num_nodes, st);
}

int ovs_bow_transform(const ovs_bow_vocab* v, const uint8_t* descs, int n, int levelsup, uint32_t* word_of_feature, uint32_t* node_of_feature,
                      uint32_t* bow_word_ids, double* bow_values, int32_t* num_words, uint32_t* fv_node_ids, uint32_t* fv_offsets,
                      uint32_t* fv_indices, int32_t* fv_num_nodes) {
    int rc = check_vocab(v);
    if (rc != OVS_OK) {
        return rc;
    }
    if (n < 0 || !num_words || !fv_num_nodes || (n > 0 && (!descs || !bow_word_ids || !bow_values || !fv_node_ids || !fv_offsets || !fv_indices))) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "invalid argument");
    }
    if (n >= (1 << 20)) {
        return ovs::fail(OVS_ERR_UNSUPPORTED, "more than 2^20 descriptors per frame are not supported");
    }
    *num_words = 0;
    *fv_num_nodes = 0;
    if (n == 0 || v->empty) {
        if (fv_offsets) {
            fv_offsets[0] = 0;
        }
        return OVS_OK;
    }
    ovs::device_guard guard(v->device);
    thread_ctx* ctx = ctx_for(v->device);
    if ((rc = ensure_stream(ctx)) != OVS_OK) {
        return rc;
    }
    const size_t nn = n;
    const size_t total = al(nn * 32) + 2 * al(nn * 4) + al(nn * 4) + al(nn * 8) + al(256) + al(nn * 4) + al((nn + 1) * 4) + al(nn * 4)
                         + bow_scratch::bytes(1, nn);
    if ((rc = ctx->arena.reserve(total)) != OVS_OK) {
        return rc;
    }
    carver cv;
    cv.base = ctx->arena.as<uint8_t>();
    auto* d_descs = cv.take<uint8_t>(nn * 32);
    auto* d_word_of = cv.take<u32>(nn);
    auto* d_node_of = cv.take<u32>(nn);
    auto* d_bow_ids = cv.take<u32>(nn);
    auto* d_bow_val = cv.take<double>(nn);
    auto* d_nums = cv.take<int32_t>(64); // [0] words, [1] nodes
    auto* d_fv_ids = cv.take<u32>(nn);
    auto* d_fv_off = cv.take<u32>(nn + 1);
    auto* d_fv_idx = cv.take<u32>(nn);
    bow_scratch s;
    s.carve(cv, 1, nn);
    host_stager hs;
    if ((rc = hs.init(ctx, ctx->stream)) != OVS_OK) {
        return rc;
    }
    hs.put(d_descs, descs, nn * 32);
    OVS_HIP(hs.upload());
    if ((rc = launch_transform(v, d_descs, nullptr, n, 1, n, levelsup, d_word_of, d_node_of, s, d_bow_ids, d_bow_val, d_nums, d_fv_ids, d_fv_off, d_fv_idx,
                               d_nums + 1, ctx->stream))
        != OVS_OK) {
        return rc;
    }
    int32_t nums[2] = {0, 0};
    if (word_of_feature) hs.get(word_of_feature, d_word_of, nn * 4);
    if (node_of_feature) hs.get(node_of_feature, d_node_of, nn * 4);
    hs.get(bow_word_ids, d_bow_ids, nn * 4);
    hs.get(bow_values, d_bow_val, nn * 8);
    hs.get(nums, d_nums, 8);
    hs.get(fv_node_ids, d_fv_ids, nn * 4);
    hs.get(fv_offsets, d_fv_off, (nn + 1) * 4);
    hs.get(fv_indices, d_fv_idx, nn * 4);
    OVS_HIP(hs.download());
    *num_words = nums[0];
    *fv_num_nodes = nums[1];
    return OVS_OK;
}

} // extern "C"
