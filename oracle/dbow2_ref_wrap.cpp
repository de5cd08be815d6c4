// TEST INFRASTRUCTURE.  C wrapper around the REFERENCE's own DBoW2::BowVector / DBoW2::FeatureVector, compiled (oracle/Makefile,
// target `ref`) from the unmodified sources where they lie: /root/reference/Thirdparty/DBoW2/DBoW2/{BowVector,FeatureVector}.cpp.
// These two translation units include no OpenCV header, so they are the one piece of the reference's hot path that builds in
// this image; everything else on the path includes OpenCV and is unbuildable here.  Output goes to
// oracle/_ref/ only; nothing of the reference is copied into this repository.  Used by tests/test_oracle_dbow2_ref.py to pin
// the oracle's restatement of addWeight / normalize / addFeature (bow_oracle.inc) against the real code.
#include <cstdint>
#include <cstring>
#include "BowVector.h"
#include "FeatureVector.h"

extern "C" {

// The body of TemplatedVocabulary::transform(features, v, fv, levelsup) after the per-feature tree descent
// (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1160-1203) for TF_IDF / TF weighting + L1 scoring: per feature
// `if(w > 0) { v.addWeight(id, w); fv.addFeature(nid, i_feature); }`, then `v.normalize(norm)` when must is set.
// Outputs: the BowVector in map order (ids ascending) and the FeatureVector flattened in map order.
int ref_bow_from_features(const uint32_t* word, const double* weight, const uint32_t* node, int n, int l1_normalize,
                          uint32_t* bow_word, double* bow_value, uint32_t* fv_node, uint32_t* fv_feature, int* n_fv)
{
    DBoW2::BowVector v;
    DBoW2::FeatureVector fv;
    for (int i = 0; i < n; i++)
        if (weight[i] > 0) { v.addWeight(word[i], weight[i]); fv.addFeature(node[i], (unsigned)i); }
    if (l1_normalize && !v.empty()) v.normalize(DBoW2::L1);
    int k = 0;
    for (DBoW2::BowVector::const_iterator it = v.begin(); it != v.end(); ++it, ++k) { bow_word[k] = it->first; bow_value[k] = it->second; }
    int m = 0;
    for (DBoW2::FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it)
        for (size_t j = 0; j < it->second.size(); j++, m++) { fv_node[m] = it->first; fv_feature[m] = it->second[j]; }
    *n_fv = m;
    return k;
}

// addIfNotExist (BINARY / IDF weighting path, TemplatedVocabulary.h:1180-1190)
int ref_bow_add_if_not_exist(const uint32_t* word, const double* weight, int n, uint32_t* bow_word, double* bow_value)
{
    DBoW2::BowVector v;
    for (int i = 0; i < n; i++) if (weight[i] > 0) v.addIfNotExist(word[i], weight[i]);
    int k = 0;
    for (DBoW2::BowVector::const_iterator it = v.begin(); it != v.end(); ++it, ++k) { bow_word[k] = it->first; bow_value[k] = it->second; }
    return k;
}

}
