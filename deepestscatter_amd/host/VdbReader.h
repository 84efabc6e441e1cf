// VdbReader.h -- reads the first FloatGrid of an OpenVDB file without OpenVDB, with the semantics of the
// reference's loader (Resources::loadVolumeBuffer, src/Util/Resources.cpp:82-143):
//
//     grids = openvdb::io::Stream(ifile).getGrids();  grid = gridPtrCast<FloatGrid>((*grids)[0]);       :87-88
//     maxDensity = max over the ACTIVE values (tools::extrema over cbeginValueOn: voxels and tiles)      :90-95
//     box = grid->evalActiveVoxelBoundingBox().expandBy(1);  size = box.max() + 1 - box.min()           :97-101
//     density[z][y][x] = narrow_cast<uint8_t>(accessor.getValue(min + (x,y,z)) / maxDensity * 255)      :127-141
//
// The file format is OpenVDB's own (a third-party dependency of the reference, pinned at 5.0.0 in Dependencies.md;
// not under /root/reference): restated here from its published layout -- io/Archive.cc (header, grid descriptors),
// tree/RootNode.h, InternalNode.h, LeafNode.h (readTopology / readBuffers), io/Compression.h (readCompressedValues,
// zip / blosc streams), util/NodeMasks.h (bit order).  Supported: file versions 220-224, the standard 5-4-3 float
// tree ("Tree_float_5_4_3", also saved as half), no / zip / blosc(lz4, blosclz-free) / active-mask compression,
// grid offsets or plain stream order, the linear transform maps.  Not supported (clear errors): other value types as
// the first grid, instanced grids, frustum transforms, blosc frames with a codec other than LZ4 or memcpy.
//
// No .vdb asset ships with the reference and none exists on the build image, so this reader has never seen a file
// written by Houdini or by OpenVDB itself: it is validated against files produced by an independent minimal writer
// (tests/_vdb.py, written from the same format description; its Blosc frames also with LZ4 streams from the system's
// real liblz4) -- tests/test_vdb.py.
#pragma once

#include <zlib.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <limits>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace DeepestScatter {
namespace vdb {

struct Coord {
    int32_t x, y, z;
    bool operator<(const Coord& o) const { return x != o.x ? x < o.x : (y != o.y ? y < o.y : z < o.z); }
};

constexpr uint32_t COMPRESS_ZIP = 1, COMPRESS_ACTIVE_MASK = 2, COMPRESS_BLOSC = 4;

// Files are untrusted.  A root child's or root tile's origin read from the file must be a multiple of 4096 within
// +-2^30: every coordinate derived from it (child origins, voxel positions, box edges) then fits an int32 without overflow.
// And the dense box the loader fills is capped per axis (the reference has no cap and would exhaust memory; 2048^3 floats
// are 34 GB).  The sanitizer build lowers the cap so that the rejection is exercised without the allocation
// (tests/sanitize/Makefile).
#ifndef CT_VDB_MAX_EDGE
#define CT_VDB_MAX_EDGE 2048
#endif
inline void checkRootOrigin(const Coord& o)
{
    const int32_t lim = 1 << 30;
    if (o.x < -lim || o.x > lim || o.y < -lim || o.y > lim || o.z < -lim || o.z > lim || ((o.x | o.y | o.z) & 4095) != 0) {
        throw std::runtime_error("vdb: root-level origin out of range or not a multiple of 4096");
    }
}

// ---- a byte cursor over the whole file ----------------------------------------------------------------
class Cursor {
public:
    explicit Cursor(std::vector<uint8_t> bytes) : buf(std::move(bytes)) {}
    size_t tell() const { return pos; }
    void seek(size_t p)
    {
        if (p > buf.size()) throw std::runtime_error("vdb: seek past the end of the file");
        pos = p;
    }
    void read(void* dst, size_t n)
    {
        if (n > buf.size() - pos) throw std::runtime_error("vdb: unexpected end of file");
        if (n) std::memcpy(dst, buf.data() + pos, n);   // (n == 0 comes with dst == nullptr: an empty vector's data())
        pos += n;
    }
    const uint8_t* take(size_t n)
    {
        if (n > buf.size() - pos) throw std::runtime_error("vdb: unexpected end of file");
        const uint8_t* p = buf.data() + pos;
        pos += n;
        return p;
    }
    template <typename T> T get()
    {
        T v;
        read(&v, sizeof v);
        return v;
    }
    std::string str()
    {
        const uint32_t n = get<uint32_t>();
        if (n > (1u << 24)) throw std::runtime_error("vdb: implausible string length");
        const uint8_t* p = take(n);
        return std::string(reinterpret_cast<const char*>(p), n);
    }

private:
    std::vector<uint8_t> buf;
    size_t pos = 0;
};

// ---- half -> float (IEEE 754 binary16) ------------------------------------------------------------------
inline float halfToFloat(uint16_t h)
{
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu, man = h & 0x3ffu, bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else { // subnormal
            int e = -1;
            do {
                e++;
                man <<= 1;
            } while (!(man & 0x400u));
            bits = sign | (uint32_t)(127 - 15 - e) << 23 | (man & 0x3ffu) << 13;
        }
    } else if (exp == 31) {
        bits = sign | 0x7f800000u | man << 13;
    } else {
        bits = sign | (exp + 127 - 15) << 23 | man << 13;
    }
    float f;
    std::memcpy(&f, &bits, 4);
    return f;
}

// ---- LZ4 block format ------------------------------------------------------------------------------------
inline void lz4Decode(const uint8_t* src, size_t srcLen, uint8_t* dst, size_t dstLen)
{
    size_t ip = 0, op = 0;
    while (ip < srcLen) {
        const uint8_t token = src[ip++];
        size_t lit = token >> 4;
        if (lit == 15) {
            uint8_t b;
            do {
                if (ip >= srcLen) throw std::runtime_error("vdb: corrupt LZ4 stream");
                b = src[ip++];
                lit += b;
            } while (b == 255);
        }
        if (lit > srcLen - ip || lit > dstLen - op) throw std::runtime_error("vdb: corrupt LZ4 stream");
        std::memcpy(dst + op, src + ip, lit);
        ip += lit;
        op += lit;
        if (ip >= srcLen) break; // the last sequence has literals only
        if (srcLen - ip < 2) throw std::runtime_error("vdb: corrupt LZ4 stream");
        const size_t offset = (size_t)src[ip] | (size_t)src[ip + 1] << 8;
        ip += 2;
        size_t len = token & 15u;
        if (len == 15) {
            uint8_t b;
            do {
                if (ip >= srcLen) throw std::runtime_error("vdb: corrupt LZ4 stream");
                b = src[ip++];
                len += b;
            } while (b == 255);
        }
        len += 4;
        if (offset == 0 || offset > op || len > dstLen - op) throw std::runtime_error("vdb: corrupt LZ4 stream");
        for (size_t i = 0; i < len; i++) { // byte by byte: matches may overlap their own output
            dst[op + i] = dst[op + i - offset];
        }
        op += len;
    }
    if (op != dstLen) throw std::runtime_error("vdb: LZ4 stream decodes to the wrong size");
}

// ---- Blosc 1.x frame (the container OpenVDB's bloscToStream writes: LZ4 codec, byte shuffle) ---------------
inline void bloscDecode(const uint8_t* src, size_t srcLen, uint8_t* dst, size_t dstLen)
{
    if (srcLen < 16) throw std::runtime_error("vdb: blosc frame too short");
    const uint8_t flags = src[2], typesize = src[3];
    uint32_t nbytes, blocksize, cbytes;
    std::memcpy(&nbytes, src + 4, 4);
    std::memcpy(&blocksize, src + 8, 4);
    std::memcpy(&cbytes, src + 12, 4);
    if (nbytes != dstLen || cbytes > srcLen) throw std::runtime_error("vdb: blosc frame sizes do not match the buffer");
    if (flags & 0x2) { // memcpyed
        if (srcLen < 16 + (size_t)nbytes) throw std::runtime_error("vdb: blosc frame too short");
        std::memcpy(dst, src + 16, nbytes);
        return;
    }
    const unsigned codec = flags >> 5;
    if (codec != 1) throw std::runtime_error("vdb: blosc codec " + std::to_string(codec) + " not supported (LZ4 only)");
    if (flags & 0x4) throw std::runtime_error("vdb: blosc bit-shuffle not supported");
    if (blocksize == 0 || typesize == 0) throw std::runtime_error("vdb: corrupt blosc header");
    const bool shuffled = (flags & 0x1) && typesize > 1;
    const bool dontSplit = (flags & 0x10) != 0;
    const uint32_t nblocks = (nbytes + blocksize - 1) / blocksize;
    if (srcLen < 16 + 4ull * nblocks) throw std::runtime_error("vdb: blosc frame too short");
    std::vector<uint8_t> tmp(blocksize);
    for (uint32_t b = 0; b < nblocks; b++) {
        uint32_t start;
        std::memcpy(&start, src + 16 + 4ull * b, 4);
        const bool leftover = (b == nblocks - 1) && (nbytes % blocksize != 0);
        const uint32_t bsize = leftover ? nbytes % blocksize : blocksize;
        const uint32_t nsplits = (!dontSplit && typesize <= 16 && bsize / typesize >= 128 && !leftover) ? typesize : 1;
        const uint32_t neblock = bsize / nsplits;
        size_t ip = start;
        uint8_t* out = shuffled ? tmp.data() : dst + (size_t)b * blocksize;
        for (uint32_t j = 0; j < nsplits; j++) {
            if (ip + 4 > srcLen) throw std::runtime_error("vdb: corrupt blosc block");
            int32_t c;
            std::memcpy(&c, src + ip, 4);
            ip += 4;
            if (c < 0 || ip + (size_t)c > srcLen) throw std::runtime_error("vdb: corrupt blosc block");
            if ((uint32_t)c == neblock) {
                std::memcpy(out + (size_t)j * neblock, src + ip, neblock);
            } else {
                lz4Decode(src + ip, (size_t)c, out + (size_t)j * neblock, neblock);
            }
            ip += (size_t)c;
        }
        if (shuffled) { // byte j of element i sits at tmp[j * n + i]; the last bsize % typesize bytes are copied as they are
            uint8_t* d = dst + (size_t)b * blocksize;
            const uint32_t n = bsize / typesize, rest = bsize - n * typesize;
            for (uint32_t i = 0; i < n; i++) {
                for (uint32_t j = 0; j < typesize; j++) {
                    d[(size_t)i * typesize + j] = tmp[(size_t)j * n + i];
                }
            }
            std::memcpy(d + (size_t)n * typesize, tmp.data() + (size_t)n * typesize, rest);
        }
    }
}

// io/Compression.h readData: raw, zip or blosc stream of `bytes` uncompressed bytes.
inline void readData(Cursor& in, uint8_t* dst, size_t bytes, uint32_t compression)
{
    if (compression & (COMPRESS_BLOSC | COMPRESS_ZIP)) {
        const int64_t n = in.get<int64_t>();
        if (n <= 0) { // stored uncompressed: -n bytes follow
            if ((size_t)(-n) != bytes) throw std::runtime_error("vdb: uncompressed chunk of the wrong size");
            in.read(dst, bytes);
            return;
        }
        const uint8_t* src = in.take((size_t)n);
        if (compression & COMPRESS_BLOSC) {
            bloscDecode(src, (size_t)n, dst, bytes);
        } else {
            uLongf outLen = (uLongf)bytes;
            if (uncompress(dst, &outLen, src, (uLong)n) != Z_OK || outLen != bytes) throw std::runtime_error("vdb: zlib stream is corrupt");
        }
        return;
    }
    in.read(dst, bytes);
}

// util/NodeMasks.h: bit n of a mask = bit (n & 63) of 64-bit word (n >> 6), stored as raw little-endian words.
struct Mask {
    std::vector<uint64_t> w;
    explicit Mask(size_t bits = 0) : w((bits + 63) / 64, 0) {}
    void load(Cursor& in) { in.read(w.data(), w.size() * 8); }
    bool on(size_t n) const { return (w[n >> 6] >> (n & 63)) & 1u; }
    size_t countOn() const
    {
        size_t c = 0;
        for (uint64_t v : w) c += (size_t)__builtin_popcountll(v);
        return c;
    }
};

// io/Compression.h readCompressedValues<float, MaskT>: `count` values of a node (count = bits of valueMask).
inline void readCompressedValues(Cursor& in, float* dst, size_t count, const Mask& valueMask, bool fromHalf, uint32_t compression,
                                 uint32_t fileVersion, float background)
{
    int8_t metadata = 6; // NO_MASK_AND_ALL_VALS
    if (fileVersion >= 222) metadata = in.get<int8_t>();
    // The inactive values are stored with sizeof(ValueT) = 4 bytes whatever the grid's storage precision: a half-float grid
    // truncates them to half PRECISION (truncateRealToHalf) but writes a float; only the value array goes through
    // HalfReader (io/Compression.h readCompressedValues / writeCompressedValues).
    auto readOne = [&]() -> float { return in.get<float>(); };
    float inactive1 = background;
    float inactive0 = (metadata == 0) ? background : -background;
    if (metadata == 2 || metadata == 4 || metadata == 5) {
        inactive0 = readOne();
        if (metadata == 5) inactive1 = readOne();
    }
    Mask selection(count);
    if (metadata == 3 || metadata == 4 || metadata == 5) selection.load(in);
    const bool maskCompressed = (compression & COMPRESS_ACTIVE_MASK) != 0;
    size_t tempCount = count;
    if (maskCompressed && metadata != 6 && fileVersion >= 222) tempCount = valueMask.countOn();
    std::vector<float> temp(tempCount);
    if (fromHalf) {
        std::vector<uint16_t> h(tempCount);
        readData(in, reinterpret_cast<uint8_t*>(h.data()), tempCount * 2, compression);
        for (size_t i = 0; i < tempCount; i++) temp[i] = halfToFloat(h[i]);
    } else {
        readData(in, reinterpret_cast<uint8_t*>(temp.data()), tempCount * 4, compression);
    }
    if (maskCompressed && tempCount != count) {
        for (size_t d = 0, t = 0; d < count; d++) {
            dst[d] = valueMask.on(d) ? temp[t++] : (selection.on(d) ? inactive1 : inactive0);
        }
    } else {
        std::copy(temp.begin(), temp.end(), dst);
    }
}

// ---- the 5-4-3 tree, kept as a map of dense 8^3 leaves plus the tiles of the upper levels ---------------------
struct Leaf {
    Coord origin;
    Mask valueMask{ 512 };
    std::array<float, 512> values{};
};
struct Tile {
    Coord origin;
    int32_t dim; // edge in voxels: 8 (level-1 tile), 128 (level-2 tile), 4096 (root tile)
    float value;
    bool active;
};

class FloatGrid {
public:
    std::string name, type;
    float background = 0.f;
    uint32_t fileVersion = 0, compression = 0;
    std::vector<std::unique_ptr<Leaf>> leaves;   // in file order
    std::vector<Tile> tiles;

    // tree::ValueAccessor::getValue: leaf voxel (active or not), else the smallest tile that contains it, else background
    float getValue(Coord c) const
    {
        const Coord lo{ c.x & ~7, c.y & ~7, c.z & ~7 };
        auto it = leafIndex.find(lo);
        if (it != leafIndex.end()) {
            return it->second->values[(size_t)((c.x & 7) << 6 | (c.y & 7) << 3 | (c.z & 7))];
        }
        const Tile* best = nullptr;
        for (const Tile& t : tiles) {
            if (c.x >= t.origin.x && c.y >= t.origin.y && c.z >= t.origin.z && c.x < t.origin.x + t.dim && c.y < t.origin.y + t.dim &&
                c.z < t.origin.z + t.dim && (!best || t.dim < best->dim)) {
                best = &t;
            }
        }
        return best ? best->value : background;
    }

    // Grid::evalActiveVoxelBoundingBox: every active voxel and every active tile; false when nothing is active
    bool activeBBox(Coord& lo, Coord& hi) const
    {
        const int32_t big = std::numeric_limits<int32_t>::max();
        lo = Coord{ big, big, big };
        hi = Coord{ -big, -big, -big };
        auto grow = [&](int32_t x0, int32_t y0, int32_t z0, int32_t x1, int32_t y1, int32_t z1) {
            lo = Coord{ std::min(lo.x, x0), std::min(lo.y, y0), std::min(lo.z, z0) };
            hi = Coord{ std::max(hi.x, x1), std::max(hi.y, y1), std::max(hi.z, z1) };
        };
        for (const auto& l : leaves) {
            for (int n = 0; n < 512; n++) {
                if (l->valueMask.on((size_t)n)) {
                    const int32_t x = l->origin.x + (n >> 6), y = l->origin.y + ((n >> 3) & 7), z = l->origin.z + (n & 7);
                    grow(x, y, z, x, y, z);
                }
            }
        }
        for (const Tile& t : tiles) {
            if (t.active) grow(t.origin.x, t.origin.y, t.origin.z, t.origin.x + t.dim - 1, t.origin.y + t.dim - 1, t.origin.z + t.dim - 1);
        }
        return lo.x <= hi.x;
    }

    // tools::extrema over cbeginValueOn: the maximum of the active voxel values and the active tile values
    double activeMax() const
    {
        double m = -std::numeric_limits<double>::infinity();
        for (const auto& l : leaves) {
            for (int n = 0; n < 512; n++) {
                if (l->valueMask.on((size_t)n)) m = std::max(m, (double)l->values[(size_t)n]);
            }
        }
        for (const Tile& t : tiles) {
            if (t.active) m = std::max(m, (double)t.value);
        }
        return m;
    }

    void index()
    {
        leafIndex.clear();
        for (const auto& l : leaves) leafIndex[l->origin] = l.get();
    }

private:
    std::map<Coord, const Leaf*> leafIndex;
};

namespace detail {

inline void skipMetaMap(Cursor& in)
{
    const uint32_t count = in.get<uint32_t>();
    for (uint32_t i = 0; i < count; i++) {
        in.str(); // name
        in.str(); // type name
        const uint32_t bytes = in.get<uint32_t>();
        in.take(bytes);
    }
}

inline void skipTransform(Cursor& in)
{
    const std::string type = in.str();
    size_t bytes;
    if (type == "UniformScaleMap" || type == "ScaleMap") bytes = 5 * 24;
    else if (type == "UniformScaleTranslateMap" || type == "ScaleTranslateMap") bytes = 6 * 24;
    else if (type == "TranslationMap") bytes = 24;
    else if (type == "AffineMap" || type == "UnitaryMap") bytes = 128;
    else throw std::runtime_error("vdb: transform map '" + type + "' not supported");
    in.take(bytes);
}

// InternalNode<.., LOG2DIM>::readTopology for the two upper levels; CHILD_TOTAL = log2 of the child's edge in voxels
template <int LOG2DIM, int CHILD_TOTAL>
void readInternal(Cursor& in, FloatGrid& g, Coord origin, bool fromHalf, std::vector<Leaf*>& leafOrder)
{
    constexpr size_t N = (size_t)1 << (3 * LOG2DIM);
    Mask childMask(N), valueMask(N);
    childMask.load(in);
    valueMask.load(in);
    std::vector<float> values(N, g.background);
    if (g.fileVersion < 222) {
        // before node-mask compression (OPENVDB_FILE_VERSION_NODE_MASK_COMPRESSION = 222) only the childMask.countOff()
        // values of the slots WITHOUT a child are stored, in slot order (InternalNode::readTopology, oldVersion branch)
        size_t off = 0;
        for (size_t n = 0; n < N; n++) off += childMask.on(n) ? 0u : 1u;
        std::vector<float> packed(off);
        readCompressedValues(in, packed.data(), off, valueMask, fromHalf, g.compression, g.fileVersion, g.background);
        for (size_t n = 0, k = 0; n < N; n++) {
            if (!childMask.on(n)) values[n] = packed[k++];
        }
    } else {
        readCompressedValues(in, values.data(), N, valueMask, fromHalf, g.compression, g.fileVersion, g.background);
    }
    constexpr int DIM = 1 << LOG2DIM;
    auto childOrigin = [&](size_t n) {
        const int32_t x = (int32_t)(n >> (2 * LOG2DIM)), y = (int32_t)((n >> LOG2DIM) & (DIM - 1)), z = (int32_t)(n & (DIM - 1));
        return Coord{ origin.x + (x << CHILD_TOTAL), origin.y + (y << CHILD_TOTAL), origin.z + (z << CHILD_TOTAL) };
    };
    for (size_t n = 0; n < N; n++) {
        if (!childMask.on(n) && (valueMask.on(n) || values[n] != g.background)) {
            g.tiles.push_back(Tile{ childOrigin(n), 1 << CHILD_TOTAL, values[n], valueMask.on(n) });
        }
    }
    for (size_t n = 0; n < N; n++) {
        if (!childMask.on(n)) continue;
        if constexpr (CHILD_TOTAL == 7) {
            readInternal<4, 3>(in, g, childOrigin(n), fromHalf, leafOrder);
        } else {
            auto leaf = std::make_unique<Leaf>(); // LeafNode::readTopology: the value mask
            leaf->origin = childOrigin(n);
            leaf->valueMask.load(in);
            leaf->values.fill(g.background);
            leafOrder.push_back(leaf.get());
            g.leaves.push_back(std::move(leaf));
        }
    }
}

} // namespace detail

// The first grid of the file, which must be a float grid (gridPtrCast<FloatGrid>((*grids)[0]), Resources.cpp:87-88).
inline FloatGrid readFirstFloatGrid(const std::string& path)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f.good()) throw std::runtime_error("cannot open " + path);
    const std::streamsize size = f.tellg();
    f.seekg(0);
    std::vector<uint8_t> bytes((size_t)size);
    f.read(reinterpret_cast<char*>(bytes.data()), size);
    Cursor in(std::move(bytes));

    // ---- io::Archive::readHeader
    if (in.get<int64_t>() != 0x56444220) throw std::runtime_error("vdb: not an OpenVDB file (magic number)");
    FloatGrid g;
    g.fileVersion = in.get<uint32_t>();
    if (g.fileVersion < 220 || g.fileVersion > 224) {
        throw std::runtime_error("vdb: file version " + std::to_string(g.fileVersion) + " not supported (220-224)");
    }
    in.get<uint32_t>(); // library major
    in.get<uint32_t>(); // library minor
    const bool hasGridOffsets = in.get<uint8_t>() != 0;
    uint32_t fileCompression = COMPRESS_ZIP | COMPRESS_ACTIVE_MASK;
    if (g.fileVersion < 222) {
        fileCompression = in.get<uint8_t>() ? (COMPRESS_ZIP | COMPRESS_ACTIVE_MASK) : 0u; // one flag for the whole file
    }
    in.take(36); // UUID, ASCII
    detail::skipMetaMap(in); // file-level metadata
    const uint32_t gridCount = in.get<uint32_t>();
    if (gridCount == 0) throw std::runtime_error("vdb: the file holds no grid");

    // ---- GridDescriptor::read
    g.name = in.str();
    g.type = in.str();
    const std::string instanceParent = in.str();
    if (!instanceParent.empty()) throw std::runtime_error("vdb: instanced grids are not supported");
    const int64_t gridPos = in.get<int64_t>();
    in.get<int64_t>(); // blockPos
    in.get<int64_t>(); // endPos
    bool fromHalf = false;
    std::string type = g.type;
    const std::string halfSuffix = "_HalfFloat";
    if (type.size() > halfSuffix.size() && type.compare(type.size() - halfSuffix.size(), halfSuffix.size(), halfSuffix) == 0) {
        fromHalf = true;
        type.resize(type.size() - halfSuffix.size());
    }
    if (type != "Tree_float_5_4_3") {
        throw std::runtime_error("vdb: first grid '" + g.name + "' has type " + g.type + ", expected a FloatGrid (Tree_float_5_4_3)");
    }
    if (hasGridOffsets) in.seek((size_t)gridPos);

    // ---- Archive::readGrid: per-grid compression flags, metadata, transform, topology, buffers
    g.compression = (g.fileVersion >= 222) ? in.get<uint32_t>() : fileCompression;
    detail::skipMetaMap(in);
    detail::skipTransform(in);

    // Tree::readTopology
    if (in.get<uint32_t>() != 1) throw std::runtime_error("vdb: multi-buffer trees are not supported");
    // RootNode::readTopology
    // (background and root tile values: sizeof(ValueType) = 4 bytes also in a half-float grid, which only truncates
    // them to half precision before writing -- RootNode::writeTopology)
    g.background = in.get<float>();
    const uint32_t numTiles = in.get<uint32_t>(), numChildren = in.get<uint32_t>();
    for (uint32_t i = 0; i < numTiles; i++) {
        Coord o;
        in.read(&o, 12);
        checkRootOrigin(o);
        const float v = in.get<float>();
        const bool active = in.get<uint8_t>() != 0;
        g.tiles.push_back(Tile{ o, 4096, v, active });
    }
    std::vector<Leaf*> leafOrder;
    for (uint32_t i = 0; i < numChildren; i++) {
        Coord o;
        in.read(&o, 12);
        checkRootOrigin(o);
        detail::readInternal<5, 7>(in, g, o, fromHalf, leafOrder);
    }
    // Tree::readBuffers -> LeafNode::readBuffers, in the order of the topology
    for (Leaf* leaf : leafOrder) {
        leaf->valueMask.load(in);
        if (g.fileVersion < 222) { // older files repeat the origin and give a buffer count (always one)
            in.take(12);
            if (in.get<int8_t>() != 1) throw std::runtime_error("vdb: multi-buffer leaves are not supported");
        }
        readCompressedValues(in, leaf->values.data(), 512, leaf->valueMask, fromHalf, g.compression, g.fileVersion, g.background);
    }
    g.index();
    return g;
}

// Resources::loadVolumeBuffer (Resources.cpp:90-141) on top of the reader: the uint8 texture, x fastest, z slowest.
// The reference fills the box voxel by voxel through a ValueAccessor (:127-141); a node of the tree is either a child
// or a tile of its parent, so pasting the tiles and then the leaves into a dense box gives the same values.
inline void loadVolumeTexture(const std::string& path, std::vector<uint8_t>& texture, std::array<uint32_t, 3>& dims)
{
    const FloatGrid g = readFirstFloatGrid(path);
    Coord lo, hi;
    if (!g.activeBBox(lo, hi)) throw std::runtime_error("vdb: grid '" + g.name + "' has no active voxel");
    const double maxDensity = g.activeMax();
    // boundingBox.expandBy(1); min = box.min(); max = box.max() + 1; size = max - min
    const Coord mn{ lo.x - 1, lo.y - 1, lo.z - 1 };
    const int64_t sx = (int64_t)hi.x + 2 - mn.x, sy = (int64_t)hi.y + 2 - mn.y, sz = (int64_t)hi.z + 2 - mn.z;
    if (sx > CT_VDB_MAX_EDGE || sy > CT_VDB_MAX_EDGE || sz > CT_VDB_MAX_EDGE) {
        throw std::runtime_error("vdb: active bounding box larger than " + std::to_string(CT_VDB_MAX_EDGE) + " voxels");
    }
    dims = { (uint32_t)sx, (uint32_t)sy, (uint32_t)sz };
    std::vector<float> dense((size_t)(sx * sy * sz), g.background); // [z][y][x]
    auto paste = [&](Coord o, int32_t dim, auto&& valueAt) {
        const int32_t x0 = std::max(o.x, mn.x), y0 = std::max(o.y, mn.y), z0 = std::max(o.z, mn.z);
        const int32_t x1 = (int32_t)std::min<int64_t>((int64_t)o.x + dim, mn.x + sx), y1 = (int32_t)std::min<int64_t>((int64_t)o.y + dim, mn.y + sy),
                      z1 = (int32_t)std::min<int64_t>((int64_t)o.z + dim, mn.z + sz);
        for (int32_t z = z0; z < z1; z++) {
            for (int32_t y = y0; y < y1; y++) {
                float* row = dense.data() + ((size_t)(z - mn.z) * (size_t)sy + (size_t)(y - mn.y)) * (size_t)sx;
                for (int32_t x = x0; x < x1; x++) {
                    row[x - mn.x] = valueAt(x - o.x, y - o.y, z - o.z);
                }
            }
        }
    };
    for (const Tile& t : g.tiles) {
        paste(t.origin, t.dim, [&](int32_t, int32_t, int32_t) { return t.value; });
    }
    for (const auto& l : g.leaves) {
        paste(l->origin, 8, [&](int32_t x, int32_t y, int32_t z) { return l->values[(size_t)(x << 6 | y << 3 | z)]; });
    }
    texture.resize(dense.size());
    for (size_t i = 0; i < dense.size(); i++) {
        texture[i] = (uint8_t)((double)dense[i] / maxDensity * 255); // narrow_cast<uint8_t>(float / double * int), :137
    }
}

} // namespace vdb
} // namespace DeepestScatter
