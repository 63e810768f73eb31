// The witness table's WitnessData column, byte-compatible with the reference:
//     WitnessData = base64.StdEncoding( s2.Encode( gob( utils.BatchCreateUserWitness ) ) )
// written by serializeWorker (src/witness/witness/witness.go:215-232) and read by utils.DecodeBatchWitness
// (src/utils/utils.go:704-742), which also expands every user's sparse asset list to the dense AssetCounts form.
//
// Three layers, each restated from its PUBLISHED format (Go toolchain and the klauspost/compress module are absent from the image):
//   * gob  — Go's encoding/gob wire format as specified in the package documentation: unsigned / signed integer encoding, byte
//            counts, type definition messages (wireType / StructType / ArrayType / SliceType / gobEncoderType values with the
//            bootstrap type ids), struct values as (field delta, value) pairs closed by 0, zero fields omitted.  The encoder emits
//            one valid stream (types defined before use, our own id numbering); the decoder accepts ANY valid stream for these
//            types — ids, definition order and omitted fields are the sender's choice — matching fields by NAME as gob does.
//            math/big.Int travels as a GobEncoder value: one byte (version 1 << 1 | sign) + big-endian magnitude.
//            Pinned by the documentation's own worked example (type Point struct{X, Y int}; Point{22, 33}), tests/test_witness_codec_cpu.py.
//   * s2   — the block format of github.com/klauspost/compress/s2 (go.mod: v1.17.10): uvarint length + Snappy elements (literal,
//            copy1, copy2, copy4), plus S2's repeat-offset extension (copy1 with offset 0) which only the DECODER needs — the
//            encoder here emits Snappy-compatible elements only (greedy hash matcher), every such block is a valid s2 block.
//            The repeat extension is restated from memory of s2's decoder: UNPINNED (no golden stream in the reference).
//   * base64 — RFC 4648 with padding (proof_row.hpp).
// Host-only, no device.  Names follow src/utils/types.go.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#include "proof_row.hpp"

namespace zkpor_host {

typedef std::string Bytes;  // raw bytes

// ------------------------------------------------------------------------------------------------ data model (types.go)
struct BigIntW {  // *big.Int: nil, or sign + big-endian magnitude without leading zeros
    bool present = false, neg = false;
    Bytes mag;
    static BigIntW from_u128(unsigned __int128 v) {
        BigIntW b;
        b.present = true;
        for (int i = 15; i >= 0; --i) {
            uint8_t byte = (uint8_t)(v >> (8 * i));
            if (byte || !b.mag.empty()) b.mag.push_back((char)byte);
        }
        return b;
    }
    bool operator==(const BigIntW& o) const { return present == o.present && neg == o.neg && mag == o.mag; }
};
struct TierRatioW { BigIntW BoundaryValue; uint8_t Ratio = 0; BigIntW PrecomputedValue; };          // types.go:5-9
static const int kTierCount = 12, kAccountTreeDepth = 28, kAssetCounts = 500;                      // constants.go:18-21
struct CexAssetInfoW {                                                                              // types.go:11-23
    uint64_t TotalEquity = 0, TotalDebt = 0, BasePrice = 0;
    std::string Symbol;
    uint32_t Index = 0;
    uint64_t LoanCollateral = 0, MarginCollateral = 0, PortfolioMarginCollateral = 0;
    std::array<TierRatioW, kTierCount> LoanRatios, MarginRatios, PortfolioMarginRatios;
};
struct AccountAssetW { uint16_t Index = 0; uint64_t Equity = 0, Debt = 0, Loan = 0, Margin = 0, PortfolioMargin = 0; };  // types.go:25-32
struct CreateUserOperationW {                                                                       // types.go:43-48
    std::vector<AccountAssetW> Assets;
    uint32_t AccountIndex = 0;
    Bytes AccountIdHash;
    std::array<Bytes, kAccountTreeDepth> AccountProof;
};
struct BatchCreateUserWitnessW {                                                                    // types.go:50-60
    Bytes BatchCommitment, AccountTreeRoot, BeforeCEXAssetsCommitment, AfterCEXAssetsCommitment;
    uint32_t MinAccountIndex = 0, MaxAccountIndex = 0;
    std::vector<CexAssetInfoW> BeforeCexAssets;
    std::vector<CreateUserOperationW> CreateUserOps;
};

// ------------------------------------------------------------------------------------------------ gob: primitives
namespace gob {

enum : int64_t { tBool = 1, tInt = 2, tUint = 3, tFloat = 4, tBytes = 5, tString = 6, tComplex = 7, tInterface = 8,
                 tWireType = 16, tArrayType = 17, tCommonType = 18, tSliceType = 19, tStructType = 20, tFieldType = 21,
                 tFieldTypeSlice = 22, tMapType = 23, tGobEncoderType = 24, firstUserId = 64 };

inline void put_uint(Bytes& o, uint64_t v) {
    if (v < 128) { o.push_back((char)v); return; }
    int n = 0;
    for (uint64_t t = v; t; t >>= 8) ++n;
    o.push_back((char)(uint8_t)(-n));
    for (int i = n - 1; i >= 0; --i) o.push_back((char)(uint8_t)(v >> (8 * i)));
}
inline void put_int(Bytes& o, int64_t i) {
    uint64_t u = i < 0 ? (~((uint64_t)i << 1)) : ((uint64_t)i << 1);
    put_uint(o, u);
}
inline void put_bytes(Bytes& o, const Bytes& b) { put_uint(o, b.size()); o += b; }

struct Reader {
    const uint8_t* p;
    size_t n, off = 0;
    Reader(const uint8_t* d, size_t len) : p(d), n(len) {}
    bool eof() const { return off >= n; }
    uint8_t byte() { if (off >= n) throw std::runtime_error("gob: truncated"); return p[off++]; }
    uint64_t uint() {
        uint8_t b = byte();
        if (b < 128) return b;
        int cnt = -(int)(int8_t)b;
        if (cnt < 1 || cnt > 8) throw std::runtime_error("gob: bad integer byte count");
        uint64_t v = 0;
        for (int i = 0; i < cnt; ++i) v = (v << 8) | byte();
        return v;
    }
    int64_t sint() {
        uint64_t u = uint();
        return (u & 1) ? (int64_t)~(u >> 1) : (int64_t)(u >> 1);
    }
    Bytes bytes() {
        uint64_t len = uint();
        if (len > n - off) throw std::runtime_error("gob: byte count beyond the message");
        Bytes b((const char*)p + off, (size_t)len);
        off += (size_t)len;
        return b;
    }
};

// ---- type descriptions (what a wireType value says) ----
struct Field { std::string name; int64_t id; };
struct WireType {
    enum Kind { Array, Slice, Struct, Map, GobEncoder } kind = Struct;
    std::string name;
    int64_t id = 0, elem = 0, key = 0, len = 0;
    std::vector<Field> fields;
};

// message framing: uint(byte length) + payload
inline void put_message(Bytes& out, const Bytes& payload) { put_uint(out, payload.size()); out += payload; }

// (-id, wireType) — the definition message of one type, exactly as the package documentation lays it out
inline Bytes type_definition(const WireType& t) {
    Bytes b;
    put_int(b, -t.id);
    auto common = [&](Bytes& o) {           // CommonType{Name, Id}
        if (!t.name.empty()) { put_uint(o, 1); put_bytes(o, t.name); put_uint(o, 1); } else put_uint(o, 2);
        put_int(o, t.id);
        put_uint(o, 0);
    };
    switch (t.kind) {
        case WireType::Array:               // wireType field 0: ArrayType{CommonType, Elem, Len}
            put_uint(b, 1); put_uint(b, 1); common(b); put_uint(b, 1); put_int(b, t.elem);
            if (t.len) { put_uint(b, 1); put_int(b, t.len); }
            put_uint(b, 0);
            break;
        case WireType::Slice:               // field 1: SliceType{CommonType, Elem}
            put_uint(b, 2); put_uint(b, 1); common(b); put_uint(b, 1); put_int(b, t.elem); put_uint(b, 0);
            break;
        case WireType::Struct:              // field 2: StructType{CommonType, Field []*fieldType{Name, Id}}
            put_uint(b, 3); put_uint(b, 1); common(b);
            if (!t.fields.empty()) {
                put_uint(b, 1); put_uint(b, t.fields.size());
                for (auto& f : t.fields) { put_uint(b, 1); put_bytes(b, f.name); put_uint(b, 1); put_int(b, f.id); put_uint(b, 0); }
            }
            put_uint(b, 0);
            break;
        case WireType::GobEncoder:          // field 4: gobEncoderType{CommonType}
            put_uint(b, 5); put_uint(b, 1); common(b); put_uint(b, 0);
            break;
        default: throw std::runtime_error("gob: map types are not used by the witness");
    }
    put_uint(b, 0);                         // end of wireType
    return b;
}

// ---- generic decoded value ----
struct Value {
    enum Kind { Nil, Uint, Int, Raw, List, Struct } kind = Nil;
    uint64_t u = 0;
    int64_t i = 0;
    Bytes raw;
    std::vector<Value> list;
    std::map<std::string, Value> fields;
    const Value* get(const std::string& k) const { auto it = fields.find(k); return it == fields.end() ? nullptr : &it->second; }
};

class Decoder {
public:
    // consumes the whole stream: definition messages, then ONE value message; returns the value and its type id
    Value decode(const uint8_t* data, size_t len, int64_t* type_id = nullptr) {
        Reader r(data, len);
        while (!r.eof()) {
            uint64_t mlen = r.uint();
            if (mlen > r.n - r.off) throw std::runtime_error("gob: message longer than the stream");
            Reader m(r.p + r.off, (size_t)mlen);
            r.off += (size_t)mlen;
            int64_t id = m.sint();
            if (id < 0) { define(-id, m); continue; }
            if (type_id) *type_id = id;
            Value v;
            if (is_struct(id)) v = value(id, m);
            else { if (m.uint() != 0) throw std::runtime_error("gob: non-struct top-level value without the 0 marker"); v = value(id, m); }
            if (!m.eof()) throw std::runtime_error("gob: bytes left in the value message");
            if (!r.eof()) throw std::runtime_error("gob: bytes after the value message");
            return v;
        }
        throw std::runtime_error("gob: no value message");
    }
    const std::map<int64_t, WireType>& types() const { return types_; }

private:
    std::map<int64_t, WireType> types_;
    bool is_struct(int64_t id) const { auto it = types_.find(id); return it != types_.end() && it->second.kind == WireType::Struct; }

    static void common(Reader& m, WireType& t) {  // CommonType value
        int f = -1;
        for (;;) {
            uint64_t d = m.uint();
            if (!d) break;
            f += (int)d;
            if (f == 0) t.name = m.bytes();
            else if (f == 1) t.id = m.sint();
            else throw std::runtime_error("gob: unknown CommonType field");
        }
    }
    void define(int64_t id, Reader& m) {
        if (id < firstUserId || types_.count(id)) throw std::runtime_error("gob: duplicate or reserved type id");
        WireType t;
        int wf = -1;
        bool seen = false;
        for (;;) {  // wireType struct: exactly one of its pointer fields is set
            uint64_t d = m.uint();
            if (!d) break;
            wf += (int)d;
            if (seen) throw std::runtime_error("gob: wireType with two descriptions");
            seen = true;
            t.kind = wf == 0 ? WireType::Array : wf == 1 ? WireType::Slice : wf == 2 ? WireType::Struct : wf == 3 ? WireType::Map : WireType::GobEncoder;
            if (wf > 6) throw std::runtime_error("gob: unknown wireType field");
            int f = -1;
            for (;;) {
                uint64_t d2 = m.uint();
                if (!d2) break;
                f += (int)d2;
                if (f == 0) { common(m, t); continue; }
                if (wf == 0) { if (f == 1) t.elem = m.sint(); else if (f == 2) t.len = m.sint(); else throw std::runtime_error("gob: ArrayType field"); }
                else if (wf == 1) { if (f == 1) t.elem = m.sint(); else throw std::runtime_error("gob: SliceType field"); }
                else if (wf == 2) {
                    if (f != 1) throw std::runtime_error("gob: StructType field");
                    uint64_t cnt = m.uint();
                    if (cnt > m.n) throw std::runtime_error("gob: field count");
                    for (uint64_t k = 0; k < cnt; ++k) {
                        Field fl{"", 0};
                        int ff = -1;
                        for (;;) {
                            uint64_t d3 = m.uint();
                            if (!d3) break;
                            ff += (int)d3;
                            if (ff == 0) fl.name = m.bytes(); else if (ff == 1) fl.id = m.sint(); else throw std::runtime_error("gob: fieldType field");
                        }
                        t.fields.push_back(fl);
                    }
                } else if (wf == 3) { if (f == 1) t.key = m.sint(); else if (f == 2) t.elem = m.sint(); else throw std::runtime_error("gob: MapType field"); }
                else throw std::runtime_error("gob: gobEncoderType field");
            }
        }
        if (!seen) throw std::runtime_error("gob: empty wireType");
        if (!m.eof()) throw std::runtime_error("gob: bytes left in a type definition");
        t.id = id;
        types_[id] = t;
    }
    Value value(int64_t id, Reader& m, int depth = 0) {
        if (depth > 64) throw std::runtime_error("gob: nesting too deep");
        Value v;
        switch (id) {
            case tBool: case tUint: v.kind = Value::Uint; v.u = m.uint(); return v;
            case tInt: v.kind = Value::Int; v.i = m.sint(); return v;
            case tBytes: case tString: v.kind = Value::Raw; v.raw = m.bytes(); return v;
            case tFloat: v.kind = Value::Uint; v.u = m.uint(); return v;
            default: break;
        }
        auto it = types_.find(id);
        if (it == types_.end()) throw std::runtime_error("gob: value of an undefined type");
        const WireType& t = it->second;
        switch (t.kind) {
            case WireType::GobEncoder: v.kind = Value::Raw; v.raw = m.bytes(); return v;
            case WireType::Array: case WireType::Slice: {
                uint64_t cnt = m.uint();
                if (t.kind == WireType::Array && (int64_t)cnt != t.len) throw std::runtime_error("gob: array length differs from its type");
                if (cnt > m.n - m.off) throw std::runtime_error("gob: element count beyond the message");  // every element is >= 1 byte
                v.kind = Value::List;
                v.list.reserve((size_t)cnt);
                for (uint64_t k = 0; k < cnt; ++k) v.list.push_back(value(t.elem, m, depth + 1));
                return v;
            }
            case WireType::Struct: {
                v.kind = Value::Struct;
                int f = -1;
                for (;;) {
                    uint64_t d = m.uint();
                    if (!d) break;
                    f += (int)d;
                    if (f < 0 || (size_t)f >= t.fields.size()) throw std::runtime_error("gob: field number beyond the struct type");
                    v.fields[t.fields[f].name] = value(t.fields[f].id, m, depth + 1);
                }
                return v;
            }
            default: throw std::runtime_error("gob: map values are not used by the witness");
        }
    }
};

}  // namespace gob

// ------------------------------------------------------------------------------------------------ gob: the witness
namespace witness_gob {
using namespace gob;
// our numbering of the user types (any consistent numbering is valid gob)
enum : int64_t { idWitness = 65, idCexSlice = 66, idCex = 67, idTierArray = 68, idTier = 69, idBigInt = 70, idOpSlice = 71, idOp = 72,
                 idAssetSlice = 73, idAsset = 74, idProofArray = 75 };

inline std::vector<WireType> types() {
    auto st = [](int64_t id, const char* name, std::vector<Field> f) { WireType t; t.kind = WireType::Struct; t.id = id; t.name = name; t.fields = std::move(f); return t; };
    auto sl = [](int64_t id, const char* name, int64_t elem) { WireType t; t.kind = WireType::Slice; t.id = id; t.name = name; t.elem = elem; return t; };
    auto ar = [](int64_t id, const char* name, int64_t elem, int64_t len) { WireType t; t.kind = WireType::Array; t.id = id; t.name = name; t.elem = elem; t.len = len; return t; };
    WireType big; big.kind = WireType::GobEncoder; big.id = idBigInt; big.name = "Int";
    return {
        st(idWitness, "BatchCreateUserWitness", {{"BatchCommitment", tBytes}, {"AccountTreeRoot", tBytes}, {"BeforeCEXAssetsCommitment", tBytes},
                                                 {"AfterCEXAssetsCommitment", tBytes}, {"MinAccountIndex", tUint}, {"MaxAccountIndex", tUint},
                                                 {"BeforeCexAssets", idCexSlice}, {"CreateUserOps", idOpSlice}}),
        sl(idCexSlice, "[]utils.CexAssetInfo", idCex),
        st(idCex, "CexAssetInfo", {{"TotalEquity", tUint}, {"TotalDebt", tUint}, {"BasePrice", tUint}, {"Symbol", tString}, {"Index", tUint},
                                   {"LoanCollateral", tUint}, {"MarginCollateral", tUint}, {"PortfolioMarginCollateral", tUint},
                                   {"LoanRatios", idTierArray}, {"MarginRatios", idTierArray}, {"PortfolioMarginRatios", idTierArray}}),
        ar(idTierArray, "[12]utils.TierRatio", idTier, kTierCount),
        st(idTier, "TierRatio", {{"BoundaryValue", idBigInt}, {"Ratio", tUint}, {"PrecomputedValue", idBigInt}}),
        big,
        sl(idOpSlice, "[]utils.CreateUserOperation", idOp),
        st(idOp, "CreateUserOperation", {{"Assets", idAssetSlice}, {"AccountIndex", tUint}, {"AccountIdHash", tBytes}, {"AccountProof", idProofArray}}),
        sl(idAssetSlice, "[]utils.AccountAsset", idAsset),
        st(idAsset, "AccountAsset", {{"Index", tUint}, {"Equity", tUint}, {"Debt", tUint}, {"Loan", tUint}, {"Margin", tUint}, {"PortfolioMargin", tUint}}),
        ar(idProofArray, "[28][]uint8", tBytes, kAccountTreeDepth),
    };
}

// struct body writer: fields in order, zero values skipped, deltas from the last field sent
struct StructW {
    Bytes& o;
    int last = -1;
    explicit StructW(Bytes& out) : o(out) {}
    void delta(int f) { put_uint(o, (uint64_t)(f - last)); last = f; }
    void u(int f, uint64_t v) { if (v) { delta(f); put_uint(o, v); } }
    void b(int f, const Bytes& v) { if (!v.empty()) { delta(f); put_bytes(o, v); } }
    void end() { put_uint(o, 0); }
};

inline Bytes bigint_gob(const BigIntW& x) {  // math/big (*Int).GobEncode: version 1, sign in bit 0, then |x| big-endian
    Bytes b;
    b.push_back((char)(uint8_t)((1 << 1) | (x.neg ? 1 : 0)));
    return b + x.mag;
}
inline void put_tier(Bytes& o, const TierRatioW& t) {
    StructW s(o);
    if (t.BoundaryValue.present) { s.delta(0); put_bytes(o, bigint_gob(t.BoundaryValue)); }
    s.u(1, t.Ratio);
    if (t.PrecomputedValue.present) { s.delta(2); put_bytes(o, bigint_gob(t.PrecomputedValue)); }
    s.end();
}
inline void put_tiers(Bytes& o, StructW& s, int f, const std::array<TierRatioW, kTierCount>& a) {
    // an array field is omitted only if every element is zero (gob's isZero on arrays)
    bool zero = true;
    for (auto& t : a) zero &= !t.BoundaryValue.present && !t.PrecomputedValue.present && t.Ratio == 0;
    if (zero) return;
    s.delta(f);
    put_uint(o, kTierCount);
    for (auto& t : a) put_tier(o, t);
}

// gob(BatchCreateUserWitness): the definition messages, then the value message
inline Bytes Encode(const BatchCreateUserWitnessW& w) {
    Bytes out;
    for (auto& t : types()) put_message(out, type_definition(t));
    Bytes v;
    put_int(v, idWitness);
    StructW s(v);
    s.b(0, w.BatchCommitment); s.b(1, w.AccountTreeRoot); s.b(2, w.BeforeCEXAssetsCommitment); s.b(3, w.AfterCEXAssetsCommitment);
    s.u(4, w.MinAccountIndex); s.u(5, w.MaxAccountIndex);
    if (!w.BeforeCexAssets.empty()) {
        s.delta(6);
        put_uint(v, w.BeforeCexAssets.size());
        for (auto& c : w.BeforeCexAssets) {
            StructW cs(v);
            cs.u(0, c.TotalEquity); cs.u(1, c.TotalDebt); cs.u(2, c.BasePrice); cs.b(3, c.Symbol); cs.u(4, c.Index);
            cs.u(5, c.LoanCollateral); cs.u(6, c.MarginCollateral); cs.u(7, c.PortfolioMarginCollateral);
            put_tiers(v, cs, 8, c.LoanRatios); put_tiers(v, cs, 9, c.MarginRatios); put_tiers(v, cs, 10, c.PortfolioMarginRatios);
            cs.end();
        }
    }
    if (!w.CreateUserOps.empty()) {
        s.delta(7);
        put_uint(v, w.CreateUserOps.size());
        for (auto& op : w.CreateUserOps) {
            StructW os(v);
            if (!op.Assets.empty()) {
                os.delta(0);
                put_uint(v, op.Assets.size());
                for (auto& a : op.Assets) {
                    StructW as(v);
                    as.u(0, a.Index); as.u(1, a.Equity); as.u(2, a.Debt); as.u(3, a.Loan); as.u(4, a.Margin); as.u(5, a.PortfolioMargin);
                    as.end();
                }
            }
            os.u(1, op.AccountIndex); os.b(2, op.AccountIdHash);
            bool any = false;
            for (auto& p : op.AccountProof) any |= !p.empty();
            if (any) {
                os.delta(3);
                put_uint(v, kAccountTreeDepth);
                for (auto& p : op.AccountProof) put_bytes(v, p);
            }
            os.end();
        }
    }
    s.end();
    put_message(out, v);
    return out;
}

inline uint64_t as_u(const Value* v, uint64_t max, const char* what) {
    if (!v) return 0;
    if (v->kind != Value::Uint || v->u > max) throw std::runtime_error(std::string("witness: field ") + what + " is not an unsigned integer in range");
    return v->u;
}
inline Bytes as_b(const Value* v, const char* what) {
    if (!v) return Bytes();
    if (v->kind != Value::Raw) throw std::runtime_error(std::string("witness: field ") + what + " is not a byte string");
    return v->raw;
}
inline BigIntW as_big(const Value* v) {
    BigIntW b;
    if (!v) return b;
    if (v->kind != Value::Raw) throw std::runtime_error("witness: big integer is not a GobEncoder value");
    b.present = true;
    if (v->raw.empty()) return b;                       // (*Int)(nil).GobEncode() is empty; decodes to 0
    uint8_t h = (uint8_t)v->raw[0];
    if ((h >> 1) != 1) throw std::runtime_error("witness: big.Int encoding version is not 1");
    b.neg = h & 1;
    b.mag = v->raw.substr(1);
    return b;
}
inline void tiers_from(const Value* v, std::array<TierRatioW, kTierCount>& out) {
    if (!v) return;
    if (v->kind != Value::List || v->list.size() != (size_t)kTierCount) throw std::runtime_error("witness: tier ratio array");
    for (int i = 0; i < kTierCount; ++i) {
        const Value& t = v->list[i];
        if (t.kind != Value::Struct) throw std::runtime_error("witness: tier ratio");
        out[i].BoundaryValue = as_big(t.get("BoundaryValue"));
        out[i].Ratio = (uint8_t)as_u(t.get("Ratio"), 255, "Ratio");
        out[i].PrecomputedValue = as_big(t.get("PrecomputedValue"));
    }
}

// any valid gob stream of a BatchCreateUserWitness (ours or Go's) -> the struct, asset lists as stored (sparse)
inline BatchCreateUserWitnessW Decode(const Bytes& stream) {
    Decoder d;
    Value v = d.decode((const uint8_t*)stream.data(), stream.size());
    if (v.kind != Value::Struct) throw std::runtime_error("witness: top-level value is not a struct");
    BatchCreateUserWitnessW w;
    w.BatchCommitment = as_b(v.get("BatchCommitment"), "BatchCommitment");
    w.AccountTreeRoot = as_b(v.get("AccountTreeRoot"), "AccountTreeRoot");
    w.BeforeCEXAssetsCommitment = as_b(v.get("BeforeCEXAssetsCommitment"), "BeforeCEXAssetsCommitment");
    w.AfterCEXAssetsCommitment = as_b(v.get("AfterCEXAssetsCommitment"), "AfterCEXAssetsCommitment");
    w.MinAccountIndex = (uint32_t)as_u(v.get("MinAccountIndex"), 0xffffffffull, "MinAccountIndex");
    w.MaxAccountIndex = (uint32_t)as_u(v.get("MaxAccountIndex"), 0xffffffffull, "MaxAccountIndex");
    if (const Value* cs = v.get("BeforeCexAssets")) {
        if (cs->kind != Value::List) throw std::runtime_error("witness: BeforeCexAssets");
        for (auto& c : cs->list) {
            if (c.kind != Value::Struct) throw std::runtime_error("witness: CexAssetInfo");
            CexAssetInfoW a;
            a.TotalEquity = as_u(c.get("TotalEquity"), ~0ull, "TotalEquity"); a.TotalDebt = as_u(c.get("TotalDebt"), ~0ull, "TotalDebt");
            a.BasePrice = as_u(c.get("BasePrice"), ~0ull, "BasePrice"); a.Symbol = as_b(c.get("Symbol"), "Symbol");
            a.Index = (uint32_t)as_u(c.get("Index"), 0xffffffffull, "Index");
            a.LoanCollateral = as_u(c.get("LoanCollateral"), ~0ull, "LoanCollateral");
            a.MarginCollateral = as_u(c.get("MarginCollateral"), ~0ull, "MarginCollateral");
            a.PortfolioMarginCollateral = as_u(c.get("PortfolioMarginCollateral"), ~0ull, "PortfolioMarginCollateral");
            tiers_from(c.get("LoanRatios"), a.LoanRatios); tiers_from(c.get("MarginRatios"), a.MarginRatios);
            tiers_from(c.get("PortfolioMarginRatios"), a.PortfolioMarginRatios);
            w.BeforeCexAssets.push_back(std::move(a));
        }
    }
    if (const Value* ops = v.get("CreateUserOps")) {
        if (ops->kind != Value::List) throw std::runtime_error("witness: CreateUserOps");
        for (auto& o : ops->list) {
            if (o.kind != Value::Struct) throw std::runtime_error("witness: CreateUserOperation");
            CreateUserOperationW op;
            if (const Value* as = o.get("Assets")) {
                if (as->kind != Value::List) throw std::runtime_error("witness: Assets");
                for (auto& a : as->list) {
                    if (a.kind != Value::Struct) throw std::runtime_error("witness: AccountAsset");
                    AccountAssetW x;
                    x.Index = (uint16_t)as_u(a.get("Index"), 0xffff, "Index"); x.Equity = as_u(a.get("Equity"), ~0ull, "Equity");
                    x.Debt = as_u(a.get("Debt"), ~0ull, "Debt"); x.Loan = as_u(a.get("Loan"), ~0ull, "Loan");
                    x.Margin = as_u(a.get("Margin"), ~0ull, "Margin"); x.PortfolioMargin = as_u(a.get("PortfolioMargin"), ~0ull, "PortfolioMargin");
                    op.Assets.push_back(x);
                }
            }
            op.AccountIndex = (uint32_t)as_u(o.get("AccountIndex"), 0xffffffffull, "AccountIndex");
            op.AccountIdHash = as_b(o.get("AccountIdHash"), "AccountIdHash");
            if (const Value* pr = o.get("AccountProof")) {
                if (pr->kind != Value::List || pr->list.size() != (size_t)kAccountTreeDepth) throw std::runtime_error("witness: AccountProof");
                for (int i = 0; i < kAccountTreeDepth; ++i) op.AccountProof[i] = as_b(&pr->list[i], "AccountProof");
            }
            w.CreateUserOps.push_back(std::move(op));
        }
    }
    return w;
}
}  // namespace witness_gob

// ------------------------------------------------------------------------------------------------ s2 block format
namespace s2 {

inline void put_uvarint(Bytes& o, uint64_t v) {
    while (v >= 0x80) { o.push_back((char)(uint8_t)(v | 0x80)); v >>= 7; }
    o.push_back((char)(uint8_t)v);
}
inline void emit_literal(Bytes& o, const uint8_t* p, size_t n) {
    while (n) {
        size_t k = n > (1u << 24) ? (1u << 24) : n;  // up to 3 length bytes per element
        size_t l = k - 1;
        if (l < 60) o.push_back((char)(uint8_t)(l << 2));
        else if (l < (1u << 8)) { o.push_back((char)(uint8_t)(60 << 2)); o.push_back((char)(uint8_t)l); }
        else if (l < (1u << 16)) { o.push_back((char)(uint8_t)(61 << 2)); o.push_back((char)(uint8_t)l); o.push_back((char)(uint8_t)(l >> 8)); }
        else { o.push_back((char)(uint8_t)(62 << 2)); o.push_back((char)(uint8_t)l); o.push_back((char)(uint8_t)(l >> 8)); o.push_back((char)(uint8_t)(l >> 16)); }
        o.append((const char*)p, k);
        p += k; n -= k;
    }
}
inline void emit_copy(Bytes& o, size_t offset, size_t len) {  // Snappy elements only: copy1 (len 4..11, offset < 2048) or copy2 (len 1..64)
    while (len) {
        if (len >= 4 && len <= 11 && offset < 2048) {
            o.push_back((char)(uint8_t)(1 | ((len - 4) << 2) | ((offset >> 8) << 5)));
            o.push_back((char)(uint8_t)offset);
            return;
        }
        size_t k = len > 64 ? 64 : len;
        if (len > 64 && len - k < 4) k = len - 4;  // leave a tail of at least 4 so it can still be a copy
        o.push_back((char)(uint8_t)(2 | ((k - 1) << 2)));
        o.push_back((char)(uint8_t)offset); o.push_back((char)(uint8_t)(offset >> 8));
        len -= k;
    }
}

// s2.Encode-compatible block: uvarint(len) + elements.  level 0 = literals only; 1 = greedy 4-byte hash matcher (offsets < 65536)
inline Bytes Encode(const Bytes& src, int level = 1) {
    Bytes o;
    put_uvarint(o, src.size());
    const uint8_t* p = (const uint8_t*)src.data();
    const size_t n = src.size();
    if (level == 0 || n < 16) { emit_literal(o, p, n); return o; }
    std::vector<uint32_t> table(1 << 14, 0xffffffffu);
    auto load = [&](size_t i) { uint32_t v; memcpy(&v, p + i, 4); return v; };
    size_t lit = 0, i = 0;
    while (i + 4 <= n) {
        uint32_t v = load(i);
        uint32_t h = (v * 0x1e35a7bdu) >> 18;
        uint32_t cand = table[h];
        table[h] = (uint32_t)i;
        if (cand != 0xffffffffu && i - cand < 65536 && load(cand) == v) {
            size_t m = 4;
            while (i + m < n && p[cand + m] == p[i + m]) ++m;
            if (i > lit) emit_literal(o, p + lit, i - lit);
            emit_copy(o, i - cand, m);
            i += m;
            lit = i;
            continue;
        }
        ++i;
    }
    if (n > lit) emit_literal(o, p + lit, n - lit);
    return o;
}

// s2.Decode of one block (Snappy elements + S2's repeat-offset form of copy1)
inline Bytes Decode(const Bytes& in) {
    const uint8_t* s = (const uint8_t*)in.data();
    const size_t n = in.size();
    size_t i = 0;
    uint64_t want = 0;
    int shift = 0;
    for (;;) {
        if (i >= n || shift > 63) throw std::runtime_error("s2: bad length prefix");
        uint8_t b = s[i++];
        want |= (uint64_t)(b & 0x7f) << shift;
        if (!(b & 0x80)) break;
        shift += 7;
    }
    if (want > (1ull << 32)) throw std::runtime_error("s2: block too large");
    Bytes out;
    out.reserve((size_t)(want < 64 * (uint64_t)n + 64 ? want : 64 * (uint64_t)n + 64));  // a damaged length prefix must not reserve gigabytes
    size_t last_offset = 1;
    auto need = [&](size_t k) { if (n - i < k) throw std::runtime_error("s2: truncated element"); };
    while (i < n) {
        uint8_t tag = s[i];
        size_t len = 0, offset = 0;
        switch (tag & 3) {
            case 0: {  // literal
                size_t l = tag >> 2;
                ++i;
                if (l >= 60) {
                    size_t extra = l - 59;
                    need(extra);
                    l = 0;
                    for (size_t k = 0; k < extra; ++k) l |= (size_t)s[i + k] << (8 * k);
                    i += extra;
                }
                ++l;
                need(l);
                if (out.size() + l > want) throw std::runtime_error("s2: output overrun");
                out.append((const char*)s + i, l);
                i += l;
                continue;
            }
            case 1: {  // copy1: 3-bit length, 11-bit offset; offset 0 = repeat the previous offset with an extended length (S2)
                need(2);
                offset = ((size_t)(tag & 0xe0) << 3) | s[i + 1];
                len = (tag >> 2) & 7;
                i += 2;
                if (offset == 0) {
                    offset = last_offset;
                    if (len == 5) { need(1); len = (size_t)s[i] + 4; i += 1; }
                    else if (len == 6) { need(2); len = ((size_t)s[i] | (size_t)s[i + 1] << 8) + (1 << 8); i += 2; }
                    else if (len == 7) { need(3); len = ((size_t)s[i] | (size_t)s[i + 1] << 8 | (size_t)s[i + 2] << 16) + (1 << 16); i += 3; }
                }
                len += 4;
                break;
            }
            case 2: need(3); len = (tag >> 2) + 1; offset = (size_t)s[i + 1] | (size_t)s[i + 2] << 8; i += 3; break;
            default: need(5); len = (tag >> 2) + 1; offset = (size_t)s[i + 1] | (size_t)s[i + 2] << 8 | (size_t)s[i + 3] << 16 | (size_t)s[i + 4] << 24; i += 5; break;
        }
        if (offset == 0 || offset > out.size() || out.size() + len > want) throw std::runtime_error("s2: bad copy");
        last_offset = offset;
        size_t from = out.size() - offset;
        for (size_t k = 0; k < len; ++k) out.push_back(out[from + k]);  // may overlap itself (run-length)
    }
    if (out.size() != want) throw std::runtime_error("s2: length mismatch");
    return out;
}
}  // namespace s2

inline Bytes base64_std_decode(const std::string& in) {
    // one table lookup per character and a sized output: a 5 MB witness column decodes in a few milliseconds
    static const std::array<int8_t, 256> T = [] {
        std::array<int8_t, 256> t;
        t.fill(-1);
        for (int i = 0; i < 26; ++i) { t['A' + i] = (int8_t)i; t['a' + i] = (int8_t)(26 + i); }
        for (int i = 0; i < 10; ++i) t['0' + i] = (int8_t)(52 + i);
        t[(unsigned char)'+'] = 62; t[(unsigned char)'/'] = 63;
        return t;
    }();
    const size_t n = in.size();
    if (n % 4) throw std::runtime_error("base64: length is not a multiple of 4");
    if (n == 0) return Bytes();
    const unsigned char* p = (const unsigned char*)in.data();
    const int pad = (p[n - 2] == '=') + (p[n - 1] == '=');
    if (p[n - 2] == '=' && p[n - 1] != '=') throw std::runtime_error("base64: bad padding");
    Bytes out(n / 4 * 3 - (size_t)pad, '\0');
    char* o = &out[0];
    const size_t full = n - (pad ? 4 : 0);                      // quads without padding
    for (size_t i = 0; i < full; i += 4) {
        const int a = T[p[i]], b = T[p[i + 1]], c = T[p[i + 2]], d = T[p[i + 3]];
        if ((a | b | c | d) < 0) {
            if (p[i] == '=' || p[i + 1] == '=' || p[i + 2] == '=' || p[i + 3] == '=') throw std::runtime_error("base64: padding inside the data");
            throw std::runtime_error("base64: bad character");
        }
        const uint32_t w = (uint32_t)a << 18 | (uint32_t)b << 12 | (uint32_t)c << 6 | (uint32_t)d;
        *o++ = (char)(uint8_t)(w >> 16); *o++ = (char)(uint8_t)(w >> 8); *o++ = (char)(uint8_t)w;
    }
    if (pad) {
        const size_t i = n - 4;
        const int a = T[p[i]], b = T[p[i + 1]], c = pad == 2 ? 0 : T[p[i + 2]];
        if ((a | b | c) < 0) throw std::runtime_error("base64: bad character");
        const uint32_t w = (uint32_t)a << 18 | (uint32_t)b << 12 | (uint32_t)c << 6;
        *o++ = (char)(uint8_t)(w >> 16);
        if (pad < 2) *o++ = (char)(uint8_t)(w >> 8);
    }
    return out;
}

// ------------------------------------------------------------------------------------------------ the column
// serializeWorker (witness.go:215-232): the string stored in witness{suffix}.WitnessData
inline std::string EncodeBatchWitness(const BatchCreateUserWitnessW& w, int s2_level = 1) {
    return base64_std(s2::Encode(witness_gob::Encode(w), s2_level));
}
// utils.DecodeBatchWitness (utils.go:704-742), including the expansion of every user's stored (sparse) asset list to the dense
// AssetCounts-entry list the circuit assignment indexes by position
inline BatchCreateUserWitnessW DecodeBatchWitness(const std::string& data, bool expand_assets = true) {
    BatchCreateUserWitnessW w = witness_gob::Decode(s2::Decode(base64_std_decode(data)));
    if (expand_assets)
        for (auto& op : w.CreateUserOps) {
            std::vector<AccountAssetW> dense(kAssetCounts);
            for (int p = 0; p < kAssetCounts; ++p) dense[p].Index = (uint16_t)p;
            for (auto& a : op.Assets) {
                if (a.Index >= kAssetCounts) throw std::runtime_error("witness: asset index beyond AssetCounts");  // Go: index out of range panic
                dense[a.Index] = a;
            }
            op.Assets = std::move(dense);
        }
    return w;
}

}  // namespace zkpor_host
