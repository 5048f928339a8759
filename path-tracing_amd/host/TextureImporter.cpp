#include "TextureImporter.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

namespace PathTracing
{

// A decoded image is at most this many pixels: 16x the 4096 x 4096 the reference keeps textures within AFTER its scaling
// (TextureUploader.h:74) -- sources may be larger than what is uploaded (skies are), but a corrupted header must not turn
// into a multi-gigabyte allocation, and every decoder checks the data that is really there before it sizes a buffer.
constexpr uint64_t kMaxImagePixels = 16384ull * 16384ull;


std::vector<uint8_t> ReadFileBytes(const std::filesystem::path &path)
{
    std::error_code ec;
    if (!std::filesystem::is_regular_file(path, ec)) // fopen() opens a directory too, and ftell() on it answers LONG_MAX
        throw error("Could not open " + path.string());
    FILE *f = std::fopen(path.string().c_str(), "rb");
    if (!f)
        throw error("Could not open " + path.string());
    std::fseek(f, 0, SEEK_END);
    const long size = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> bytes(size > 0 ? static_cast<size_t>(size) : 0);
    const bool ok = bytes.empty() || std::fread(bytes.data(), 1, bytes.size(), f) == bytes.size();
    std::fclose(f);
    if (!ok)
        throw error("Could not read " + path.string());
    return bytes;
}

// =======================================================================================================
// inflate (RFC 1951) inside a zlib wrapper (RFC 1950)
// =======================================================================================================

namespace
{

struct BitReader
{
    const uint8_t *p, *end;
    uint32_t acc = 0;
    int n = 0;
    uint32_t Bits(int count)
    {
        while (n < count)
        {
            if (p >= end)
                throw error("inflate: unexpected end of data");
            acc |= static_cast<uint32_t>(*p++) << n;
            n += 8;
        }
        const uint32_t v = acc & ((count == 32) ? 0xffffffffu : ((1u << count) - 1u));
        acc >>= count;
        n -= count;
        return v;
    }
    void AlignToByte()
    {
        acc = 0;
        n = 0;
    }
};

// canonical Huffman table decoded bit by bit (the counting method of RFC 1951 3.2.2)
struct Huffman
{
    uint16_t count[16] = {};
    uint16_t symbol[288] = {};
    void Build(const uint8_t *lengths, int n)
    {
        std::memset(count, 0, sizeof(count));
        for (int i = 0; i < n; i++)
            count[lengths[i]]++;
        count[0] = 0;
        uint16_t offs[16];
        offs[1] = 0;
        for (int len = 1; len < 15; len++)
            offs[len + 1] = static_cast<uint16_t>(offs[len] + count[len]);
        for (int i = 0; i < n; i++)
            if (lengths[i])
                symbol[offs[lengths[i]]++] = static_cast<uint16_t>(i);
    }
    int Decode(BitReader &br) const
    {
        int code = 0, first = 0, index = 0;
        for (int len = 1; len <= 15; len++)
        {
            code |= static_cast<int>(br.Bits(1));
            const int c = count[len];
            if (code - c < first)
                return symbol[index + (code - first)];
            index += c;
            first += c;
            first <<= 1;
            code <<= 1;
        }
        throw error("inflate: bad Huffman code");
    }
};

const uint16_t kLenBase[29] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
const uint8_t kLenExtra[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
const uint16_t kDistBase[30] = { 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577 };
const uint8_t kDistExtra[30] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };

void InflateBlock(BitReader &br, const Huffman &lit, const Huffman &dist, std::vector<uint8_t> &out)
{
    for (;;)
    {
        const int sym = lit.Decode(br);
        if (sym < 256)
            out.push_back(static_cast<uint8_t>(sym));
        else if (sym == 256)
            return;
        else
        {
            if (sym > 285)
                throw error("inflate: bad length symbol");
            const size_t len = kLenBase[sym - 257] + br.Bits(kLenExtra[sym - 257]);
            const int ds = dist.Decode(br);
            if (ds > 29)
                throw error("inflate: bad distance symbol");
            const size_t d = kDistBase[ds] + br.Bits(kDistExtra[ds]);
            if (d > out.size())
                throw error("inflate: distance beyond the window");
            const size_t start = out.size() - d;
            for (size_t i = 0; i < len; i++)
                out.push_back(out[start + i]);
        }
    }
}

}

std::vector<uint8_t> TextureImporter::Inflate(std::span<const uint8_t> z)
{
    if (z.size() < 6 || (z[0] & 0x0f) != 8 || ((z[0] << 8) | z[1]) % 31 != 0 || (z[1] & 0x20))
        throw error("inflate: not a zlib stream");
    BitReader br { z.data() + 2, z.data() + z.size() };
    std::vector<uint8_t> out;
    bool last = false;
    while (!last)
    {
        last = br.Bits(1) != 0;
        const uint32_t type = br.Bits(2);
        if (type == 0)
        {
            br.AlignToByte();
            if (br.end - br.p < 4)
                throw error("inflate: truncated stored block");
            const uint32_t len = br.p[0] | (br.p[1] << 8), nlen = br.p[2] | (br.p[3] << 8);
            br.p += 4;
            if ((len ^ 0xffffu) != nlen || static_cast<size_t>(br.end - br.p) < len)
                throw error("inflate: bad stored block");
            out.insert(out.end(), br.p, br.p + len);
            br.p += len;
        }
        else if (type == 1)
        {
            uint8_t lengths[288];
            for (int i = 0; i < 144; i++) lengths[i] = 8;
            for (int i = 144; i < 256; i++) lengths[i] = 9;
            for (int i = 256; i < 280; i++) lengths[i] = 7;
            for (int i = 280; i < 288; i++) lengths[i] = 8;
            Huffman lit, dist;
            lit.Build(lengths, 288);
            uint8_t dl[30];
            std::memset(dl, 5, sizeof(dl));
            dist.Build(dl, 30);
            InflateBlock(br, lit, dist, out);
        }
        else if (type == 2)
        {
            const int hlit = static_cast<int>(br.Bits(5)) + 257, hdist = static_cast<int>(br.Bits(5)) + 1, hclen = static_cast<int>(br.Bits(4)) + 4;
            static const uint8_t order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
            uint8_t cl[19] = {};
            for (int i = 0; i < hclen; i++)
                cl[order[i]] = static_cast<uint8_t>(br.Bits(3));
            Huffman clh;
            clh.Build(cl, 19);
            uint8_t lengths[320] = {};
            int i = 0;
            while (i < hlit + hdist)
            {
                const int sym = clh.Decode(br);
                if (sym < 16)
                    lengths[i++] = static_cast<uint8_t>(sym);
                else
                {
                    int rep;
                    uint8_t val = 0;
                    if (sym == 16)
                    {
                        if (i == 0)
                            throw error("inflate: repeat without a previous length");
                        val = lengths[i - 1];
                        rep = 3 + static_cast<int>(br.Bits(2));
                    }
                    else if (sym == 17)
                        rep = 3 + static_cast<int>(br.Bits(3));
                    else
                        rep = 11 + static_cast<int>(br.Bits(7));
                    if (i + rep > hlit + hdist)
                        throw error("inflate: too many code lengths");
                    while (rep--)
                        lengths[i++] = val;
                }
            }
            Huffman lit, dist;
            lit.Build(lengths, hlit);
            dist.Build(lengths + hlit, hdist);
            InflateBlock(br, lit, dist, out);
        }
        else
            throw error("inflate: reserved block type");
    }
    return out;
}

// =======================================================================================================
// PNG
// =======================================================================================================

DecodedImage TextureImporter::DecodePng(std::span<const uint8_t> f)
{
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
    if (f.size() < 33 || std::memcmp(f.data(), sig, 8) != 0)
        throw error("Not a PNG file");
    auto be32 = [&](size_t o) { return (uint32_t(f[o]) << 24) | (uint32_t(f[o + 1]) << 16) | (uint32_t(f[o + 2]) << 8) | f[o + 3]; };
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, palette, trns;
    size_t pos = 8;
    bool end = false;
    while (!end && pos + 12 <= f.size())
    {
        const uint32_t len = be32(pos);
        const char *type = reinterpret_cast<const char *>(&f[pos + 4]);
        if (pos + 12 + len > f.size())
            throw error("PNG: truncated chunk");
        const uint8_t *body = &f[pos + 8];
        if (!std::memcmp(type, "IHDR", 4))
        {
            w = be32(pos + 8);
            h = be32(pos + 12);
            depth = body[8];
            ctype = body[9];
            interlace = body[12];
        }
        else if (!std::memcmp(type, "PLTE", 4))
            palette.assign(body, body + len);
        else if (!std::memcmp(type, "tRNS", 4))
            trns.assign(body, body + len);
        else if (!std::memcmp(type, "IDAT", 4))
            idat.insert(idat.end(), body, body + len);
        else if (!std::memcmp(type, "IEND", 4))
            end = true;
        pos += 12 + len;
    }
    if (!w || !h)
        throw error("PNG: missing IHDR");
    if (static_cast<uint64_t>(w) * h > kMaxImagePixels)
        throw error("PNG: image too large");
    if (interlace > 1)
        throw error("PNG: unknown interlace method");
    const int samples = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!samples || !(depth == 8 || depth == 16 || (depth < 8 && (ctype == 0 || ctype == 3))))
        throw error("PNG: unsupported colour type / bit depth");
    const size_t bpp = std::max<size_t>(1, static_cast<size_t>(samples) * depth / 8);
    std::vector<uint8_t> raw = Inflate(idat);
    // The image data is one pass (the whole image) or the seven reduced images of Adam7 (PNG spec 8.2), each a sequence of
    // filtered scanlines of its own width; a pass of zero width or height is absent from the stream.
    struct Pass { uint32_t x0, y0, dx, dy; };
    static const Pass adam7[7] = { { 0, 0, 8, 8 }, { 4, 0, 8, 8 }, { 0, 4, 4, 8 }, { 2, 0, 4, 4 }, { 0, 2, 2, 4 }, { 1, 0, 2, 2 }, { 0, 1, 1, 2 } };
    static const Pass whole = { 0, 0, 1, 1 };
    const Pass *passes = interlace ? adam7 : &whole;
    const int passCount = interlace ? 7 : 1;
    DecodedImage img;
    img.Width = w;
    img.Height = h;
    img.Channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 4 ? 2 : ctype == 6 ? 4 : (trns.empty() ? 3 : 4);
    if ((ctype == 0 || ctype == 2) && !trns.empty())
        img.Channels++; // a colour key adds an alpha channel
    {
        // bounds before allocation: what the passes need must be in the inflated stream before w * h * 4 bytes are asked for
        // (a 100-byte file can declare 16384 x 16384)
        uint64_t needed = 0;
        for (int pass = 0; pass < passCount; pass++)
        {
            const Pass &ps = passes[pass];
            const uint64_t pw = w > ps.x0 ? (w - ps.x0 + ps.dx - 1) / ps.dx : 0, ph = h > ps.y0 ? (h - ps.y0 + ps.dy - 1) / ps.dy : 0;
            if (pw && ph)
                needed += ((pw * samples * depth + 7) / 8 + 1) * ph;
        }
        if (raw.size() < needed)
            throw error("PNG: not enough image data");
    }
    img.Pixels.resize(static_cast<size_t>(w) * h * 4);
    size_t at = 0; // position in the inflated stream
    for (int pass = 0; pass < passCount; pass++)
    {
    const Pass &ps = passes[pass];
    const uint32_t pw = w > ps.x0 ? (w - ps.x0 + ps.dx - 1) / ps.dx : 0, ph = h > ps.y0 ? (h - ps.y0 + ps.dy - 1) / ps.dy : 0;
    if (!pw || !ph)
        continue;
    const size_t stride = (static_cast<size_t>(pw) * samples * depth + 7) / 8;
    if (raw.size() < at || raw.size() - at < (stride + 1) * ph)
        throw error("PNG: not enough image data");
    // unfilter in place (PNG spec 9.2)
    for (uint32_t y = 0; y < ph; y++)
    {
        uint8_t *cur = &raw[at + (stride + 1) * y + 1];
        const uint8_t *up = y ? &raw[at + (stride + 1) * (y - 1) + 1] : nullptr;
        const int filter = raw[at + (stride + 1) * y];
        for (size_t i = 0; i < stride; i++)
        {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
            int pred = 0;
            switch (filter)
            {
            case 0: pred = 0; break;
            case 1: pred = a; break;
            case 2: pred = b; break;
            case 3: pred = (a + b) >> 1; break;
            case 4:
            {
                const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
                pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                break;
            }
            default: throw error("PNG: bad filter type");
            }
            cur[i] = static_cast<uint8_t>(cur[i] + pred);
        }
    }
    for (uint32_t py = 0; py < ph; py++)
    {
        const uint8_t *row = &raw[at + (stride + 1) * py + 1];
        for (uint32_t x = 0; x < pw; x++) // x: the pixel's column inside the pass
        {
            uint8_t *o = &img.Pixels[((static_cast<size_t>(ps.y0) + static_cast<size_t>(py) * ps.dy) * w + ps.x0 + static_cast<size_t>(x) * ps.dx) * 4];
            auto sample = [&](int k) -> uint32_t { // k-th sample of pixel x at the file's bit depth
                if (depth == 8)
                    return row[static_cast<size_t>(x) * samples + k];
                if (depth == 16)
                    return (uint32_t(row[(static_cast<size_t>(x) * samples + k) * 2]) << 8) | row[(static_cast<size_t>(x) * samples + k) * 2 + 1];
                const size_t bit = static_cast<size_t>(x) * depth;
                return (row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1u);
            };
            auto to8 = [&](uint32_t v) -> uint8_t {
                if (depth == 8) return static_cast<uint8_t>(v);
                if (depth == 16) return static_cast<uint8_t>(v >> 8);
                return static_cast<uint8_t>(v * 255u / ((1u << depth) - 1u));
            };
            if (ctype == 3)
            {
                const uint32_t idx = sample(0);
                if (idx * 3 + 2 >= palette.size())
                    throw error("PNG: palette index out of range");
                o[0] = palette[idx * 3]; o[1] = palette[idx * 3 + 1]; o[2] = palette[idx * 3 + 2];
                o[3] = idx < trns.size() ? trns[idx] : 255;
            }
            else if (ctype == 0 || ctype == 4)
            {
                const uint32_t g = sample(0);
                o[0] = o[1] = o[2] = to8(g);
                o[3] = ctype == 4 ? to8(sample(1)) : 255;
                if (ctype == 0 && trns.size() >= 2 && g == ((uint32_t(trns[0]) << 8) | trns[1]))
                    o[3] = 0;
            }
            else
            {
                const uint32_t r = sample(0), g = sample(1), b = sample(2);
                o[0] = to8(r); o[1] = to8(g); o[2] = to8(b);
                o[3] = ctype == 6 ? to8(sample(3)) : 255;
                if (ctype == 2 && trns.size() >= 6 && r == ((uint32_t(trns[0]) << 8) | trns[1]) && g == ((uint32_t(trns[2]) << 8) | trns[3]) &&
                    b == ((uint32_t(trns[4]) << 8) | trns[5]))
                    o[3] = 0;
            }
        }
    }
    at += (stride + 1) * ph;
    }
    return img;
}

// =======================================================================================================
// TGA
// =======================================================================================================

DecodedImage TextureImporter::DecodeTga(std::span<const uint8_t> f)
{
    if (f.size() < 18)
        throw error("TGA: truncated header");
    const int idLen = f[0], cmapType = f[1], type = f[2], bpp = f[16], desc = f[17];
    const uint32_t w = f[12] | (f[13] << 8), h = f[14] | (f[15] << 8);
    const bool rle = type == 10 || type == 11, grey = type == 3 || type == 11;
    if (cmapType != 0 || !(type == 2 || type == 3 || type == 10 || type == 11) || !w || !h)
        throw error("TGA: unsupported image type");
    if (!((grey && bpp == 8) || (!grey && (bpp == 24 || bpp == 32))))
        throw error("TGA: unsupported pixel depth");
    const size_t px = bpp / 8;
    size_t pos = 18 + static_cast<size_t>(idLen);
    // every pixel costs at least 1 / 128 packet byte (an RLE packet covers up to 128): a header that promises more is corrupt
    if (static_cast<uint64_t>(w) * h > kMaxImagePixels || static_cast<uint64_t>(w) * h / 128 > f.size())
        throw error("TGA: image too large for its data");
    DecodedImage img;
    img.Width = w;
    img.Height = h;
    img.Channels = grey ? 1 : (bpp == 32 ? 4 : 3);
    img.Pixels.resize(static_cast<size_t>(w) * h * 4);
    const bool topDown = (desc & 0x20) != 0;
    size_t i = 0;
    const size_t total = static_cast<size_t>(w) * h;
    auto put = [&](const uint8_t *p) {
        const size_t y = i / w, x = i % w, yy = topDown ? y : h - 1 - y;
        uint8_t *o = &img.Pixels[(yy * w + x) * 4];
        if (grey) { o[0] = o[1] = o[2] = p[0]; o[3] = 255; }
        else { o[0] = p[2]; o[1] = p[1]; o[2] = p[0]; o[3] = px == 4 ? p[3] : 255; }
        i++;
    };
    while (i < total)
    {
        if (!rle)
        {
            if (pos + px > f.size()) throw error("TGA: truncated data");
            put(&f[pos]);
            pos += px;
            continue;
        }
        if (pos >= f.size()) throw error("TGA: truncated data");
        const int hdr = f[pos++], count = (hdr & 0x7f) + 1;
        if (hdr & 0x80)
        {
            if (pos + px > f.size()) throw error("TGA: truncated data");
            for (int k = 0; k < count && i < total; k++)
                put(&f[pos]);
            pos += px;
        }
        else
            for (int k = 0; k < count && i < total; k++)
            {
                if (pos + px > f.size()) throw error("TGA: truncated data");
                put(&f[pos]);
                pos += px;
            }
    }
    return img;
}

// =======================================================================================================
// Radiance HDR
// =======================================================================================================

DecodedImage TextureImporter::DecodeHdr(std::span<const uint8_t> f)
{
    const std::string_view all(reinterpret_cast<const char *>(f.data()), f.size());
    if (!(all.starts_with("#?RADIANCE") || all.starts_with("#?RGBE")))
        throw error("Not a Radiance HDR file");
    size_t pos = all.find("\n\n");
    if (pos == std::string_view::npos || all.substr(0, pos).find("FORMAT=32-bit_rle_rgbe") == std::string_view::npos)
        throw error("HDR: unsupported format");
    pos += 2;
    const size_t eol = all.find('\n', pos);
    if (eol == std::string_view::npos)
        throw error("HDR: missing resolution line");
    int w = 0, h = 0;
    if (std::sscanf(std::string(all.substr(pos, eol - pos)).c_str(), "-Y %d +X %d", &h, &w) != 2 || w <= 0 || h <= 0)
        throw error("HDR: unsupported orientation");
    pos = eol + 1;
    if (static_cast<uint64_t>(w) * static_cast<uint64_t>(h) > kMaxImagePixels || static_cast<uint64_t>(h) * 4 > all.size())
        throw error("HDR: image too large for its data"); // a scanline takes at least 4 bytes
    DecodedImage img;
    img.Width = static_cast<uint32_t>(w);
    img.Height = static_cast<uint32_t>(h);
    img.Channels = 3;
    img.IsFloat = true;
    img.Pixels.resize(static_cast<size_t>(w) * h * 16);
    float *out = reinterpret_cast<float *>(img.Pixels.data());
    std::vector<uint8_t> line(static_cast<size_t>(w) * 4);
    for (int y = 0; y < h; y++)
    {
        if (pos + 4 > f.size())
            throw error("HDR: truncated data");
        if (w >= 8 && w < 32768 && f[pos] == 2 && f[pos + 1] == 2 && ((f[pos + 2] << 8) | f[pos + 3]) == w)
        {
            pos += 4;
            for (int c = 0; c < 4; c++) // new run-length encoding: the four channels are stored one after another
            {
                int x = 0;
                while (x < w)
                {
                    if (pos >= f.size()) throw error("HDR: truncated data");
                    int count = f[pos++];
                    if (count > 128)
                    {
                        count -= 128;
                        if (pos >= f.size() || x + count > w) throw error("HDR: bad run");
                        const uint8_t v = f[pos++];
                        while (count--) line[static_cast<size_t>(x++) * 4 + c] = v;
                    }
                    else
                    {
                        if (!count || pos + count > f.size() || x + count > w) throw error("HDR: bad run");
                        while (count--) line[static_cast<size_t>(x++) * 4 + c] = f[pos++];
                    }
                }
            }
        }
        else
        {
            if (pos + static_cast<size_t>(w) * 4 > f.size()) throw error("HDR: truncated data");
            std::memcpy(line.data(), &f[pos], static_cast<size_t>(w) * 4);
            pos += static_cast<size_t>(w) * 4;
        }
        for (int x = 0; x < w; x++)
        {
            const uint8_t *p = &line[static_cast<size_t>(x) * 4];
            float *o = &out[(static_cast<size_t>(y) * w + x) * 4];
            const float s = p[3] ? std::ldexp(1.0f, static_cast<int>(p[3]) - 136) : 0.0f;
            o[0] = p[0] * s; o[1] = p[1] * s; o[2] = p[2] * s; o[3] = 1.0f;
        }
    }
    return img;
}

// =======================================================================================================
// DDS: BC1 / BC3 / BC5
// =======================================================================================================

namespace
{

void DecodeBc1Colors(const uint8_t *b, uint8_t out[16][4], bool allowPunchThrough)
{
    const uint32_t c0 = b[0] | (b[1] << 8), c1 = b[2] | (b[3] << 8);
    uint8_t pal[4][4];
    auto expand = [](uint32_t c, uint8_t *o) {
        const uint32_t r = (c >> 11) & 31, g = (c >> 5) & 63, bl = c & 31;
        o[0] = static_cast<uint8_t>((r << 3) | (r >> 2)); o[1] = static_cast<uint8_t>((g << 2) | (g >> 4)); o[2] = static_cast<uint8_t>((bl << 3) | (bl >> 2));
        o[3] = 255;
    };
    expand(c0, pal[0]);
    expand(c1, pal[1]);
    if (c0 > c1 || !allowPunchThrough)
        for (int k = 0; k < 3; k++)
        {
            pal[2][k] = static_cast<uint8_t>((2 * pal[0][k] + pal[1][k]) / 3);
            pal[3][k] = static_cast<uint8_t>((pal[0][k] + 2 * pal[1][k]) / 3);
        }
    else
        for (int k = 0; k < 3; k++)
        {
            pal[2][k] = static_cast<uint8_t>((pal[0][k] + pal[1][k]) / 2);
            pal[3][k] = 0;
        }
    pal[2][3] = 255;
    pal[3][3] = (c0 > c1 || !allowPunchThrough) ? 255 : 0;
    const uint32_t idx = b[4] | (b[5] << 8) | (b[6] << 16) | (uint32_t(b[7]) << 24);
    for (int i = 0; i < 16; i++)
        std::memcpy(out[i], pal[(idx >> (2 * i)) & 3], 4);
}

void DecodeBc4Channel(const uint8_t *b, uint8_t out[16]) // the alpha block of BC3, each half of BC5
{
    uint8_t pal[8];
    pal[0] = b[0];
    pal[1] = b[1];
    if (pal[0] > pal[1])
        for (int i = 1; i < 7; i++)
            pal[i + 1] = static_cast<uint8_t>(((7 - i) * pal[0] + i * pal[1]) / 7);
    else
    {
        for (int i = 1; i < 5; i++)
            pal[i + 1] = static_cast<uint8_t>(((5 - i) * pal[0] + i * pal[1]) / 5);
        pal[6] = 0;
        pal[7] = 255;
    }
    uint64_t bits = 0;
    for (int i = 0; i < 6; i++)
        bits |= static_cast<uint64_t>(b[2 + i]) << (8 * i);
    for (int i = 0; i < 16; i++)
        out[i] = pal[(bits >> (3 * i)) & 7];
}

}

DecodedImage TextureImporter::DecodeDds(std::span<const uint8_t> f)
{
    if (f.size() < 128 || std::memcmp(f.data(), "DDS ", 4) != 0)
        throw error("Not a DDS texture");
    auto le32 = [&](size_t o) { return uint32_t(f[o]) | (uint32_t(f[o + 1]) << 8) | (uint32_t(f[o + 2]) << 16) | (uint32_t(f[o + 3]) << 24); };
    const uint32_t h = le32(12), w = le32(16), pfFlags = le32(80), fourCC = le32(84);
    size_t offset = 128;
    enum { BC1, BC3, BC5 } fmt;
    auto cc = [](const char *s) { return uint32_t(s[0]) | (uint32_t(s[1]) << 8) | (uint32_t(s[2]) << 16) | (uint32_t(s[3]) << 24); };
    if (!(pfFlags & 4))
        throw error("Unsupported texture format");
    if (fourCC == cc("DXT1")) fmt = BC1;
    else if (fourCC == cc("DXT5")) fmt = BC3;
    else if (fourCC == cc("ATI2") || fourCC == cc("BC5U")) fmt = BC5;
    else if (fourCC == cc("DX10"))
    {
        if (f.size() < 148) throw error("DDS: truncated DX10 header");
        const uint32_t dxgi = le32(128);
        offset = 148;
        if (dxgi == 71 || dxgi == 72) fmt = BC1;
        else if (dxgi == 77 || dxgi == 78) fmt = BC3;
        else if (dxgi == 83) fmt = BC5;
        else throw error("Unsupported texture format");
    }
    else
        throw error("Unsupported texture format");
    if (!w || !h)
        throw error("DDS: empty image");
    if (static_cast<uint64_t>(w) * h > kMaxImagePixels)
        throw error("DDS: image too large");
    // the file's own mip chain (dwMipMapCount, valid with DDSD_MIPMAPCOUNT): the reference uploads it as it is
    // (TextureImporter.cpp GetTextureInfo: Levels = texture.levels(); TextureUploader.cpp:440,492-501)
    uint32_t levels = (le32(8) & 0x20000u) ? le32(28) : 1u;
    if (levels == 0)
        levels = 1;
    {
        uint32_t m = w > h ? w : h, full = 1;
        while (m > 1) { m >>= 1; full++; }
        if (levels > full)
            throw error("DDS: more mip levels than the image has");
    }
    const size_t blockBytes = fmt == BC1 ? 8 : 16;
    DecodedImage img;
    img.Width = w;
    img.Height = h;
    img.Levels = levels;
    img.Channels = fmt == BC5 ? 2 : 4;
    size_t texels = 0;
    for (uint32_t l = 0; l < levels; l++)
        texels += static_cast<size_t>(std::max(w >> l, 1u)) * std::max(h >> l, 1u);
    img.Pixels.resize(texels * 4);
    size_t out = 0;
    for (uint32_t l = 0; l < levels; l++)
    {
        const uint32_t lw = std::max(w >> l, 1u), lh = std::max(h >> l, 1u);
        const uint32_t bw = (lw + 3) / 4, bh = (lh + 3) / 4;
        if (offset + static_cast<size_t>(bw) * bh * blockBytes > f.size())
            throw error("DDS: truncated data");
        for (uint32_t by = 0; by < bh; by++)
            for (uint32_t bx = 0; bx < bw; bx++)
            {
                const uint8_t *b = &f[offset + (static_cast<size_t>(by) * bw + bx) * blockBytes];
                uint8_t px[16][4];
                if (fmt == BC1)
                    DecodeBc1Colors(b, px, true);
                else if (fmt == BC3)
                {
                    uint8_t alpha[16];
                    DecodeBc4Channel(b, alpha);
                    DecodeBc1Colors(b + 8, px, false);
                    for (int i = 0; i < 16; i++) px[i][3] = alpha[i];
                }
                else
                {
                    uint8_t r[16], g[16];
                    DecodeBc4Channel(b, r);
                    DecodeBc4Channel(b + 8, g);
                    for (int i = 0; i < 16; i++) { px[i][0] = r[i]; px[i][1] = g[i]; px[i][2] = 0; px[i][3] = 255; }
                }
                for (int i = 0; i < 16; i++)
                {
                    const uint32_t x = bx * 4 + (i & 3), y = by * 4 + (i >> 2);
                    if (x < lw && y < lh)
                        std::memcpy(&img.Pixels[(out + static_cast<size_t>(y) * lw + x) * 4], px[i], 4);
                }
            }
        offset += static_cast<size_t>(bw) * bh * blockBytes;
        out += static_cast<size_t>(lw) * lh;
    }
    return img;
}

// =======================================================================================================
// sniffing + TextureInfo
// =======================================================================================================

DecodedImage TextureImporter::Decode(std::span<const uint8_t> f)
{
    if (f.size() >= 8 && f[0] == 0x89 && f[1] == 'P' && f[2] == 'N' && f[3] == 'G')
        return DecodePng(f);
    if (f.size() >= 3 && f[0] == 0xff && f[1] == 0xd8 && f[2] == 0xff)
        return DecodeJpeg(f);
    if (f.size() >= 4 && !std::memcmp(f.data(), "DDS ", 4))
        return DecodeDds(f);
    if (f.size() >= 10 && (!std::memcmp(f.data(), "#?RADIANCE", 10) || !std::memcmp(f.data(), "#?RGBE", 6)))
        return DecodeHdr(f);
    return DecodeTga(f); // TGA has no magic number: last resort, like stb_image
}

namespace
{
TextureInfo ToTextureInfo(DecodedImage &&img, TextureType type, std::string &&name, bool *hasTransparency)
{
    if (hasTransparency)
        *hasTransparency = img.Channels == 4; // TextureImporter.cpp:300-301
    TextureInfo info;
    info.Type = type;
    info.Width = img.Width;
    info.Height = img.Height;
    info.Name = std::move(name);
    info.Format = img.IsFloat ? TextureFormat::RGBAF32 : TextureFormat::RGBAU8;
    info.Levels = img.Levels;
    info.Pixels = std::move(img.Pixels);
    if (type == TextureType::Color && img.Channels == 4 && !img.IsFloat) // PremultiplyTextureData, :24-51
        for (size_t i = 0; i + 3 < info.Pixels.size(); i += 4)
            if (info.Pixels[i + 3] == 0)
                info.Pixels[i] = info.Pixels[i + 1] = info.Pixels[i + 2] = 0;
    return info;
}
}

TextureInfo TextureImporter::GetTextureInfo(const std::filesystem::path &path, TextureType type, std::string &&name, bool *hasTransparency)
{
    const std::vector<uint8_t> bytes = ReadFileBytes(path);
    return ToTextureInfo(Decode(bytes), type, std::move(name), hasTransparency);
}

TextureInfo TextureImporter::GetTextureInfo(std::span<const uint8_t> memory, TextureType type, std::string &&name, bool *hasTransparency)
{
    return ToTextureInfo(Decode(memory), type, std::move(name), hasTransparency);
}

}
