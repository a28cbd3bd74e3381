// Integer paths of the hot path, bit-exact with the reference:
//   cdr_overlap_remap  CrossDomainDataset.calculate_user_item_from_both_domain + _remap_fields
//                      (recbole_cdr/data/dataset.py:344-445, :109-123) for ONE field (users or items): set algebra on
//                      raw tokens, overlap / target-only / source-only each sorted in Python str order (== UTF-8 byte
//                      order), ids [1,OV) overlap, [OV,OV+TO) target-only, [OV+TO,total) source-only, '[PAD]' = 0.
//   cdr_revoke_map     CrossDomainFullSortEvalDataLoader._set_user_property (recbole_cdr/data/dataloader.py:244-245):
//                      iid < OI ? iid : iid - num_target_only_item   (device kernel, int64)
// cdr_overlap_remap is host code (strings); nothing here touches the GPU except cdr_revoke_map.
#include <algorithm>
#include <cstring>
#include <string>
#include <string_view>
#include <unordered_map>
#include <unordered_set>
#include <vector>
#include "cdr_common.h"

namespace {

struct Side {
    std::vector<std::string_view> tok;   // per occurrence; empty view with data()==nullptr marks NaN
};

Side make_side(const char* bytes, const int64_t* off, const uint8_t* isnan, int64_t n) {
    Side s;
    s.tok.resize((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        if (isnan && isnan[i]) s.tok[(size_t)i] = std::string_view();
        else s.tok[(size_t)i] = std::string_view(bytes + off[i], (size_t)(off[i + 1] - off[i]));
    }
    return s;
}

__global__ void revoke_map_kernel(const int64_t* __restrict__ ids, int64_t n, int64_t OI, int64_t TOI, int64_t* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        const int64_t v = ids[e];
        out[e] = v < OI ? v : v - TOI;
    }
}

}  // namespace

extern "C" int cdr_overlap_remap(const char* src_bytes, const int64_t* src_off, const uint8_t* src_isnan, int64_t n_src,
                                 const char* tgt_bytes, const int64_t* tgt_off, const uint8_t* tgt_isnan, int64_t n_tgt,
                                 int64_t* src_ids, int64_t* tgt_ids, int64_t* counts4) {
    CDR_CHECK_ARG(src_off && tgt_off && src_ids && tgt_ids && counts4 && n_src >= 0 && n_tgt >= 0);
    CDR_CHECK_ARG((n_src == 0 || src_bytes) && (n_tgt == 0 || tgt_bytes));
    const Side S = make_side(src_bytes, src_off, src_isnan, n_src);
    const Side T = make_side(tgt_bytes, tgt_off, tgt_isnan, n_tgt);
    std::unordered_set<std::string_view> sset, tset;
    sset.reserve((size_t)n_src); tset.reserve((size_t)n_tgt);
    for (auto& t : S.tok) if (t.data()) sset.insert(t);
    for (auto& t : T.tok) if (t.data()) tset.insert(t);
    std::vector<std::string_view> overlap, s_only, t_only;
    for (auto& t : sset) (tset.count(t) ? overlap : s_only).push_back(t);
    for (auto& t : tset) if (!sset.count(t)) t_only.push_back(t);
    // Python's str ordering is code-point order; for valid UTF-8 that equals unsigned byte order
    auto lt = [](std::string_view a, std::string_view b) {                 // memcmp compares as unsigned char
        const size_t m = a.size() < b.size() ? a.size() : b.size();
        const int c = m ? memcmp(a.data(), b.data(), m) : 0;
        return c != 0 ? c < 0 : a.size() < b.size();
    };
    std::sort(overlap.begin(), overlap.end(), lt);
    std::sort(s_only.begin(), s_only.end(), lt);
    std::sort(t_only.begin(), t_only.end(), lt);
    const int64_t n_ov = (int64_t)overlap.size() + 1;     // PAD counted (dataset.py:384)
    const int64_t n_to = (int64_t)t_only.size(), n_so = (int64_t)s_only.size();
    std::unordered_map<std::string_view, int64_t> smap, tmap;
    smap.reserve(sset.size() * 2); tmap.reserve(tset.size() * 2);
    for (int64_t i = 0; i < (int64_t)overlap.size(); ++i) { smap[overlap[(size_t)i]] = i + 1; tmap[overlap[(size_t)i]] = i + 1; }
    for (int64_t i = 0; i < n_to; ++i) tmap[t_only[(size_t)i]] = n_ov + i;                 // target-only first
    for (int64_t i = 0; i < n_so; ++i) smap[s_only[(size_t)i]] = n_ov + n_to + i;
    // overlap_remap_dict['[PAD]'] = 0 is assigned after the zip (dataset.py:391), so a literal '[PAD]' token maps to 0
    const std::string_view pad("[PAD]");
    if (smap.count(pad) && tmap.count(pad) && smap[pad] < n_ov) { smap[pad] = 0; tmap[pad] = 0; }
    for (int64_t i = 0; i < n_src; ++i) src_ids[i] = S.tok[(size_t)i].data() ? smap[S.tok[(size_t)i]] : -1;
    for (int64_t i = 0; i < n_tgt; ++i) tgt_ids[i] = T.tok[(size_t)i].data() ? tmap[T.tok[(size_t)i]] : -1;
    counts4[0] = n_ov; counts4[1] = n_so; counts4[2] = n_to; counts4[3] = n_ov + n_so + n_to;
    return CDR_OK;
}

extern "C" int cdr_revoke_map(void* stream, const int64_t* ids, int64_t n, int64_t overlap_item_num,
                              int64_t target_only_item_num, int64_t* out) {
    CDR_CHECK_ARG(ids && out && n > 0);
    int64_t g = (n + 255) / 256;
    if (g > CDR_NUM_CU * 8) g = CDR_NUM_CU * 8;
    revoke_map_kernel<<<dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream>>>(ids, n, overlap_item_num, target_only_item_num, out);
    CDR_LAUNCH_CHECK();
    return CDR_OK;
}
