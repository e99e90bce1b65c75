// ovs_mapio.hip -- host-only: reads the keyframe descriptors out of a map database saved by
// io::map_database_io::save_message_pack (src/openvslam/io/map_database_io.cc:19-46), to fill the
// descriptor database of the brute-force search from a real saved map.
//
// The file is one MessagePack value (nlohmann::json::to_msgpack of the object built at
// map_database_io.cc:28-33).  Only this part of it is interpreted:
//   root["keyframes"]            object, key = std::to_string(keyframe id)  (data/map_database.cc:334-345)
//   keyframe["descs"]            array of n_keypts arrays of 8 unsigned 32-bit words: the descriptor
//                                row read as uint32_t[8] (data/common.cc:71-82, keyframe.cc:127)
// Everything else (cameras, landmarks, poses, keypoints, graph) is skipped structurally.  Rows are
// delivered in ascending keyframe id order.
#include "ovs_internal.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

namespace {

struct reader {
    const uint8_t* p;
    const uint8_t* end;
    bool ok = true;

    bool need(size_t n) {
        if (static_cast<size_t>(end - p) < n) {
            ok = false;
        }
        return ok;
    }
    uint64_t be(int n) {
        if (!need(static_cast<size_t>(n))) {
            return 0;
        }
        uint64_t v = 0;
        for (int i = 0; i < n; ++i) {
            v = (v << 8) | p[i];
        }
        p += n;
        return v;
    }
    uint8_t peek() { return need(1) ? *p : 0; }
};

enum class kind { nil, boolean, uint, sint, real, str, bin, array, map, ext, bad };

// Decodes the head of the next value.  For str / bin / ext `len` is the payload length (the payload
// itself is not consumed), for array / map the element / pair count; for uint / sint `num` holds it.
struct head {
    kind k = kind::bad;
    uint64_t len = 0;
    uint64_t num = 0;
};

head read_head(reader& r) {
    head h;
    if (!r.need(1)) {
        return h;
    }
    const uint8_t t = *r.p++;
    if (t <= 0x7f) { h.k = kind::uint; h.num = t; }
    else if (t <= 0x8f) { h.k = kind::map; h.len = t & 0x0f; }
    else if (t <= 0x9f) { h.k = kind::array; h.len = t & 0x0f; }
    else if (t <= 0xbf) { h.k = kind::str; h.len = t & 0x1f; }
    else if (t >= 0xe0) { h.k = kind::sint; h.num = static_cast<uint64_t>(static_cast<int64_t>(static_cast<int8_t>(t))); }
    else {
        switch (t) {
            case 0xc0: h.k = kind::nil; break;
            case 0xc2: case 0xc3: h.k = kind::boolean; h.num = t & 1; break;
            case 0xc4: h.k = kind::bin; h.len = r.be(1); break;
            case 0xc5: h.k = kind::bin; h.len = r.be(2); break;
            case 0xc6: h.k = kind::bin; h.len = r.be(4); break;
            case 0xc7: h.k = kind::ext; h.len = r.be(1) + 1; break;
            case 0xc8: h.k = kind::ext; h.len = r.be(2) + 1; break;
            case 0xc9: h.k = kind::ext; h.len = r.be(4) + 1; break;
            case 0xca: h.k = kind::real; h.len = 4; break;
            case 0xcb: h.k = kind::real; h.len = 8; break;
            case 0xcc: h.k = kind::uint; h.num = r.be(1); break;
            case 0xcd: h.k = kind::uint; h.num = r.be(2); break;
            case 0xce: h.k = kind::uint; h.num = r.be(4); break;
            case 0xcf: h.k = kind::uint; h.num = r.be(8); break;
            case 0xd0: h.k = kind::sint; h.num = static_cast<uint64_t>(static_cast<int64_t>(static_cast<int8_t>(r.be(1)))); break;
            case 0xd1: h.k = kind::sint; h.num = static_cast<uint64_t>(static_cast<int64_t>(static_cast<int16_t>(r.be(2)))); break;
            case 0xd2: h.k = kind::sint; h.num = static_cast<uint64_t>(static_cast<int64_t>(static_cast<int32_t>(r.be(4)))); break;
            case 0xd3: h.k = kind::sint; h.num = r.be(8); break;
            case 0xd4: h.k = kind::ext; h.len = 2; break;
            case 0xd5: h.k = kind::ext; h.len = 3; break;
            case 0xd6: h.k = kind::ext; h.len = 5; break;
            case 0xd7: h.k = kind::ext; h.len = 9; break;
            case 0xd8: h.k = kind::ext; h.len = 17; break;
            case 0xd9: h.k = kind::str; h.len = r.be(1); break;
            case 0xda: h.k = kind::str; h.len = r.be(2); break;
            case 0xdb: h.k = kind::str; h.len = r.be(4); break;
            case 0xdc: h.k = kind::array; h.len = r.be(2); break;
            case 0xdd: h.k = kind::array; h.len = r.be(4); break;
            case 0xde: h.k = kind::map; h.len = r.be(2); break;
            case 0xdf: h.k = kind::map; h.len = r.be(4); break;
            default: h.k = kind::bad; break; // 0xc1 is never used
        }
    }
    if (!r.ok) {
        h.k = kind::bad;
    }
    return h;
}

// Skips one value (iteratively: `pending` counts the values still to be consumed).
bool skip_value(reader& r) {
    uint64_t pending = 1;
    while (pending > 0 && r.ok) {
        --pending;
        const head h = read_head(r);
        switch (h.k) {
            case kind::bad: return false;
            case kind::str: case kind::bin: case kind::ext: case kind::real:
                if (!r.need(h.len)) {
                    return false;
                }
                r.p += h.len;
                break;
            case kind::array: pending += h.len; break;
            case kind::map: pending += 2 * h.len; break;
            default: break;
        }
        if (pending > static_cast<uint64_t>(r.end - r.p) + 1) {
            return false; // more values announced than bytes left
        }
    }
    return r.ok;
}

bool read_key(reader& r, std::string& key) {
    const head h = read_head(r);
    if (h.k != kind::str || !r.need(h.len)) {
        return false;
    }
    key.assign(reinterpret_cast<const char*>(r.p), h.len);
    r.p += h.len;
    return true;
}

struct keyframe_rows {
    int32_t id;
    std::vector<uint8_t> descs; // n x 32
};

bool read_descs(reader& r, std::vector<uint8_t>& out) {
    const head rows = read_head(r);
    if (rows.k != kind::array) {
        return false;
    }
    if (rows.len > static_cast<uint64_t>(r.end - r.p)) {
        return false;
    }
    out.resize(rows.len * 32);
    for (uint64_t i = 0; i < rows.len; ++i) {
        const head row = read_head(r);
        if (row.k != kind::array || row.len != 8) {
            return false;
        }
        for (int j = 0; j < 8; ++j) {
            const head w = read_head(r);
            if (w.k != kind::uint || w.num > 0xFFFFFFFFull) {
                return false;
            }
            const uint32_t v = static_cast<uint32_t>(w.num);
            std::memcpy(out.data() + 32 * i + 4 * j, &v, 4); // the word as it lies in the descriptor row
        }
    }
    return r.ok;
}

} // namespace

struct ovs_map_file {
    std::vector<keyframe_rows> keyframes; // ascending id
    int64_t total_rows = 0;
};

extern "C" {

int ovs_map_open(const char* path, ovs_map_file** out) {
    if (!path || !out) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "invalid argument");
    }
    std::FILE* f = std::fopen(path, "rb");
    if (!f) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, std::string("cannot load the file at ") + path); // map_database_io.cc:60-63
    }
    std::vector<uint8_t> bytes;
    uint8_t buf[1 << 16];
    size_t got;
    while ((got = std::fread(buf, 1, sizeof(buf), f)) > 0) {
        bytes.insert(bytes.end(), buf, buf + got);
    }
    std::fclose(f);
    reader r{bytes.data(), bytes.data() + bytes.size()};
    const head root = read_head(r);
    if (root.k != kind::map) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "not a map database: the MessagePack root is not an object");
    }
    ovs_map_file* m = new (std::nothrow) ovs_map_file;
    if (!m) {
        return ovs::fail(OVS_ERR_HIP, "out of host memory");
    }
    bool found = false, good = true;
    std::string key;
    for (uint64_t i = 0; i < root.len && good; ++i) {
        good = read_key(r, key);
        if (!good) {
            break;
        }
        if (key != "keyframes") {
            good = skip_value(r);
            continue;
        }
        found = true;
        const head kfs = read_head(r);
        if (kfs.k != kind::map) {
            good = false;
            break;
        }
        for (uint64_t k = 0; k < kfs.len && good; ++k) {
            std::string id;
            good = read_key(r, id);
            if (!good) {
                break;
            }
            // std::stoi(key), data/map_database.cc:148
            char* tail = nullptr;
            const long idv = std::strtol(id.c_str(), &tail, 10);
            if (id.empty() || *tail != '\0' || idv < 0 || idv > 0x7FFFFFFF) {
                good = false;
                break;
            }
            const head kf = read_head(r);
            if (kf.k != kind::map) {
                good = false;
                break;
            }
            keyframe_rows rows;
            rows.id = static_cast<int32_t>(idv);
            bool has_descs = false;
            for (uint64_t e = 0; e < kf.len && good; ++e) {
                good = read_key(r, key);
                if (!good) {
                    break;
                }
                if (key == "descs") {
                    good = read_descs(r, rows.descs);
                    has_descs = good;
                }
                else {
                    good = skip_value(r);
                }
            }
            good = good && has_descs;
            if (good) {
                m->keyframes.push_back(std::move(rows));
            }
        }
    }
    if (!good || !r.ok || !found) {
        delete m;
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, found ? "malformed map database (keyframes / descs)" : "map database without a \"keyframes\" object");
    }
    std::sort(m->keyframes.begin(), m->keyframes.end(), [](const keyframe_rows& a, const keyframe_rows& b) { return a.id < b.id; });
    for (const auto& kf : m->keyframes) {
        m->total_rows += static_cast<int64_t>(kf.descs.size() / 32);
    }
    *out = m;
    return OVS_OK;
}

void ovs_map_close(ovs_map_file* m) { delete m; }

int ovs_map_num_keyframes(const ovs_map_file* m) { return m ? static_cast<int>(m->keyframes.size()) : 0; }

int64_t ovs_map_num_descriptors(const ovs_map_file* m) { return m ? m->total_rows : 0; }

int ovs_map_keyframe_rows(const ovs_map_file* m, int32_t* ids, int64_t* row_offsets) {
    if (!m || !ids || !row_offsets) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "invalid argument");
    }
    int64_t off = 0;
    for (size_t i = 0; i < m->keyframes.size(); ++i) {
        ids[i] = m->keyframes[i].id;
        row_offsets[i] = off;
        off += static_cast<int64_t>(m->keyframes[i].descs.size() / 32);
    }
    row_offsets[m->keyframes.size()] = off;
    return OVS_OK;
}

int ovs_map_copy_descriptors(const ovs_map_file* m, uint8_t* descs) {
    if (!m || (m->total_rows > 0 && !descs)) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "invalid argument");
    }
    for (const auto& kf : m->keyframes) {
        if (!kf.descs.empty()) {
            std::memcpy(descs, kf.descs.data(), kf.descs.size());
            descs += kf.descs.size();
        }
    }
    return OVS_OK;
}

} // extern "C"
