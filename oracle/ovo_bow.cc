// ovo_bow.cc -- CPU ORACLE (test infrastructure; never linked into or loaded by the product).
//
// frame::compute_bow / keyframe::compute_bow (src/openvslam/data/frame.cc:162-170,
// data/keyframe.cc:181-189) call DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>::transform(
// features, bow_vec, bow_feat_vec, 4) (data/bow_vocabulary.h:16).  DBoW2 is a third-party dependency that
// is NOT in the reference tree (shinsumicco/DBoW2@687fcb7, Dockerfile.desktop:144), and neither is a
// vocabulary file (orb_vocab.dbow2 is downloaded, .travis.yml:199-206); no test of the reference holds a
// BowVector or FeatureVector.  PARITY UNPINNED: what follows restates DBoW2's published algorithm
// (TemplatedVocabulary.h: transform(features, v, fv, levelsup), transform(feature, id, weight, nid,
// levelsup); BowVector.cpp: addWeight, addIfNotExist, normalize; FeatureVector.cpp: addFeature;
// ScoringObject.cpp: mustNormalize) with the same containers (std::map) and the same double arithmetic
// in the same order, and is checked by first-principles tests only.
#include "ovo_internal.h"

#include <cmath>
#include <map>
#include <vector>

namespace {

// compute_descriptor_distance_32 (src/openvslam/match/base.h:18-39) == DBoW2::FORB::distance (bit count of
// the xor over 32 bytes)
unsigned hamming32(const uint8_t* a, const uint8_t* b) {
    unsigned d = 0;
    for (int i = 0; i < 32; ++i) {
        d += static_cast<unsigned>(__builtin_popcount(static_cast<unsigned>(a[i] ^ b[i])));
    }
    return d;
}

} // namespace

extern "C" int ovo_bow_transform(const ovs_bow_vocab_view* voc, const uint8_t* descs, int n, int levelsup, uint32_t* word_of_feature,
                                 uint32_t* node_of_feature, uint32_t* bow_word_ids, double* bow_values, int32_t* num_words,
                                 uint32_t* fv_node_ids, uint32_t* fv_offsets, uint32_t* fv_indices, int32_t* fv_num_nodes) {
    if (!voc || n < 0 || !num_words || !fv_num_nodes) {
        return ovo::fail(OVS_ERR_INVALID_ARGUMENT, "invalid argument");
    }
    *num_words = 0;
    *fv_num_nodes = 0;
    if (fv_offsets) {
        fv_offsets[0] = 0;
    }
    const uint32_t* co = voc->child_offsets;
    // transform(features, v, fv, levelsup): v.clear(); fv.clear(); if (empty()) return;
    if (voc->num_nodes <= 1 || co[1] == co[0]) {
        return OVS_OK;
    }
    std::map<uint32_t, double> v;                  // BowVector
    std::map<uint32_t, std::vector<uint32_t>> fv;  // FeatureVector
    const bool tf = voc->weighting == OVS_BOW_TF || voc->weighting == OVS_BOW_TF_IDF;
    // ScoringObject::mustNormalize: every scoring type but the dot product; L2 only for L2_NORM
    const bool must = voc->scoring != OVS_BOW_DOT_PRODUCT;
    const bool l2 = voc->scoring == OVS_BOW_L2_NORM;
    for (int i_feature = 0; i_feature < n; ++i_feature) {
        const uint8_t* feature = descs + 32 * static_cast<size_t>(i_feature);
        // transform(feature, word_id, weight, nid, levelsup)
        const int nid_level = voc->depth_levels - levelsup;
        uint32_t nid = 0; // "if (nid_level <= 0 && nid != NULL) *nid = 0; // root"
        bool nid_set = nid_level <= 0;
        uint32_t final_id = 0; // root
        int current_level = 0;
        do {
            ++current_level;
            const uint32_t* nodes = voc->children + co[final_id];
            const uint32_t count = co[final_id + 1] - co[final_id];
            final_id = nodes[0];
            double best_d = static_cast<double>(hamming32(feature, voc->node_descs + 32 * static_cast<size_t>(final_id)));
            for (uint32_t k = 1; k < count; ++k) {
                const uint32_t id = nodes[k];
                const double d = static_cast<double>(hamming32(feature, voc->node_descs + 32 * static_cast<size_t>(id)));
                if (d < best_d) {
                    best_d = d;
                    final_id = id;
                }
            }
            if (current_level == nid_level) {
                nid = final_id;
                nid_set = true;
            }
        } while (co[final_id + 1] != co[final_id]); // !m_nodes[final_id].isLeaf()
        if (!nid_set) {
            // DBoW2 leaves *nid unset when the leaf lies above nid_level; the library's rule: the leaf
            nid = final_id;
        }
        const uint32_t word_id = voc->word_ids[final_id];
        const double w = voc->weights[final_id];
        if (word_of_feature) {
            word_of_feature[i_feature] = w > 0 ? word_id : 0xFFFFFFFFu;
        }
        if (node_of_feature) {
            node_of_feature[i_feature] = nid;
        }
        if (w > 0) { // not stopped
            if (tf) {
                // BowVector::addWeight
                auto vit = v.lower_bound(word_id);
                if (vit != v.end() && !(v.key_comp()(word_id, vit->first))) {
                    vit->second += w;
                }
                else {
                    v.insert(vit, std::make_pair(word_id, w));
                }
            }
            else {
                // BowVector::addIfNotExist
                auto vit = v.lower_bound(word_id);
                if (vit == v.end() || v.key_comp()(word_id, vit->first)) {
                    v.insert(vit, std::make_pair(word_id, w));
                }
            }
            fv[nid].push_back(static_cast<uint32_t>(i_feature)); // FeatureVector::addFeature
        }
    }
    if (tf && !v.empty() && !must) {
        // unnecessary when normalizing
        const double nd = static_cast<double>(v.size());
        for (auto& kv : v) {
            kv.second /= nd;
        }
    }
    if (must) {
        // BowVector::normalize
        double norm = 0.0;
        if (!l2) {
            for (const auto& kv : v) {
                norm += std::fabs(kv.second);
            }
        }
        else {
            for (const auto& kv : v) {
                norm += kv.second * kv.second;
            }
            norm = std::sqrt(norm);
        }
        if (norm > 0.0) {
            for (auto& kv : v) {
                kv.second /= norm;
            }
        }
    }
    int k = 0;
    for (const auto& kv : v) {
        bow_word_ids[k] = kv.first;
        bow_values[k] = kv.second;
        ++k;
    }
    *num_words = k;
    int node = 0;
    uint32_t pos = 0;
    for (const auto& kv : fv) {
        fv_node_ids[node] = kv.first;
        fv_offsets[node] = pos;
        for (uint32_t idx : kv.second) {
            fv_indices[pos++] = idx;
        }
        ++node;
    }
    fv_offsets[node] = pos;
    *fv_num_nodes = node;
    return OVS_OK;
}
