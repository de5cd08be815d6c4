// Host side of the vocabulary behind the C ABI (included by sd_api.hip): the text loader of the vendored DBoW2
// (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1338-1424, FORB::fromString FORB.cpp:120-135) and the packed buffer
// that lives in HBM and travels over RCCL (k_bow.h).
#pragma once
#include <fstream>
#include <sstream>
#include <string>
#include <vector>
#include "k_bow.h"

struct sd_vocab {
    SdVocabHeader h = {};
    void* d_blob = nullptr;        // packed buffer in HBM
    bool owned = false;            // false: adopted from the caller (a broadcast target)
    SdVocabDev dev = {};
};

struct SdVocabLines {              // the node lines of the text file, in file order (node id = index + 1)
    int k = 0, L = 0, scoring = 0, weighting = 0;
    std::vector<int> parent; std::vector<uint8_t> isLeaf; std::vector<uint8_t> desc; std::vector<double> weight;
};

static size_t vocab_layout(SdVocabHeader& h)
{
    const uint64_t n = h.nNodes;
    auto al = [](uint64_t x) { return (x + 255) & ~(uint64_t)255; };
    uint64_t off = sizeof(SdVocabHeader);
    off = al(off); h.offDesc = off; off += n * 32;
    off = al(off); h.offWeight = off; off += n * 8;
    off = al(off); h.offParent = off; off += n * 4;
    off = al(off); h.offChildStart = off; off += (n + 1) * 4;
    off = al(off); h.offChildIdx = off; off += (n > 0 ? n - 1 : 0) * 4;
    off = al(off); h.offWordId = off; off += n * 4;
    h.totalBytes = al(off);
    return (size_t)h.totalBytes;
}

// Builds the packed image on the host.  Returns false when a line names a parent that does not exist yet (the reference
// would index out of bounds there).
static bool vocab_pack(const SdVocabLines& in, std::vector<uint8_t>& blob)
{
    SdVocabHeader h = {};
    const size_t lines = in.parent.size();
    h.magic = SD_VOCAB_MAGIC; h.version = 1; h.k = (uint32_t)in.k; h.L = (uint32_t)in.L; h.scoring = (uint32_t)in.scoring; h.weighting = (uint32_t)in.weighting;
    h.nNodes = (uint32_t)(lines + 1);
    const size_t bytes = vocab_layout(h);
    blob.assign(bytes, 0);
    uint8_t* desc = blob.data() + h.offDesc;
    double* weight = (double*)(blob.data() + h.offWeight);
    int* parent = (int*)(blob.data() + h.offParent);
    int* childStart = (int*)(blob.data() + h.offChildStart);
    int* childIdx = (int*)(blob.data() + h.offChildIdx);
    int* wordId = (int*)(blob.data() + h.offWordId);
    const int n = (int)h.nNodes;
    std::vector<int> nch(n, 0);
    for (size_t i = 0; i < lines; i++) {
        const int nid = (int)i + 1, pid = in.parent[i];
        if (pid < 0 || pid >= nid) return false;
        parent[nid] = pid; nch[pid]++;
        memcpy(desc + (size_t)nid * 32, in.desc.data() + i * 32, 32);
        weight[nid] = in.weight[i];
    }
    childStart[0] = 0;
    for (int i = 0; i < n; i++) childStart[i + 1] = childStart[i] + nch[i];
    std::vector<int> fill(n, 0);
    int words = 0;
    for (int i = 0; i < n; i++) wordId[i] = -1;
    for (size_t i = 0; i < lines; i++) {
        const int nid = (int)i + 1, pid = in.parent[i];
        childIdx[childStart[pid] + fill[pid]++] = nid;                 // children keep file order (m_nodes[pid].children.push_back)
        if (in.isLeaf[i] > 0) wordId[nid] = words++;                    // m_words grows in file order
    }
    h.nWords = (uint32_t)words;
    memcpy(blob.data(), &h, sizeof(h));
    return true;
}

static bool vocab_parse_text(const char* path, SdVocabLines& out, std::string& err)
{
    std::ifstream f;
    f.open(path);
    if (!f.is_open()) { err = "cannot open vocabulary file"; return false; }
    std::string s;
    getline(f, s);
    std::stringstream ss;
    ss << s;
    int n1 = -1, n2 = -1;
    out.k = -1; out.L = -1;
    ss >> out.k; ss >> out.L; ss >> n1; ss >> n2;
    if (out.k < 0 || out.k > 20 || out.L < 1 || out.L > 10 || n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3) {
        err = "Vocabulary loading failure: This is not a correct text file!";          // the reference's message (:1361)
        return false;
    }
    out.scoring = n1; out.weighting = n2;
    while (!f.eof()) {
        std::string snode;
        getline(f, snode);
        std::stringstream ssnode;
        ssnode << snode;
        int pid = 0;
        if (!(ssnode >> pid)) continue;
        int nIsLeaf = 0;
        ssnode >> nIsLeaf;
        uint8_t d[32] = {0};
        for (int iD = 0; iD < 32; iD++) { int n; if (ssnode >> n) d[iD] = (unsigned char)n; }
        double w = 0;
        ssnode >> w;
        out.parent.push_back(pid); out.isLeaf.push_back(nIsLeaf > 0 ? 1 : 0);
        out.desc.insert(out.desc.end(), d, d + 32);
        out.weight.push_back(w);
    }
    return true;
}

static void vocab_bind(sd_vocab* v)
{
    const uint8_t* p = (const uint8_t*)v->d_blob;
    v->dev.desc = p + v->h.offDesc; v->dev.weight = (const double*)(p + v->h.offWeight);
    v->dev.childStart = (const int*)(p + v->h.offChildStart); v->dev.childIdx = (const int*)(p + v->h.offChildIdx);
    v->dev.wordId = (const int*)(p + v->h.offWordId);
    v->dev.L = (int)v->h.L; v->dev.scoring = (int)v->h.scoring; v->dev.weighting = (int)v->h.weighting; v->dev.nNodes = (int)v->h.nNodes;
}
