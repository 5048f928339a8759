#include "OutputSaver.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "../../include/ptx.h"

namespace PathTracing
{

namespace
{

// ---- zlib / deflate ---------------------------------------------------------------------------------

struct BitWriter
{
    std::vector<uint8_t> &out;
    uint32_t acc = 0;
    int n = 0;
    void Bits(uint32_t value, int count) // LSB first (deflate's packing of non-Huffman fields)
    {
        acc |= value << n;
        n += count;
        while (n >= 8)
        {
            out.push_back(static_cast<uint8_t>(acc));
            acc >>= 8;
            n -= 8;
        }
    }
    void Code(uint32_t code, int count) // Huffman codes go in MSB first
    {
        uint32_t r = 0;
        for (int i = 0; i < count; i++)
            r |= ((code >> i) & 1u) << (count - 1 - i);
        Bits(r, count);
    }
    void Flush()
    {
        if (n)
            Bits(0, 8 - n);
    }
};

// fixed Huffman code of RFC 1951 3.2.6
void PutLiteralOrLength(BitWriter &bw, uint32_t symbol)
{
    if (symbol < 144)
        bw.Code(0x30 + symbol, 8);
    else if (symbol < 256)
        bw.Code(0x190 + symbol - 144, 9);
    else if (symbol < 280)
        bw.Code(symbol - 256, 7);
    else
        bw.Code(0xc0 + symbol - 280, 8);
}

const uint16_t kLengthBase[29] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
const uint8_t kLengthExtra[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
const uint16_t kDistBase[30] = { 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577 };
const uint8_t kDistExtra[30] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };

// One fixed-Huffman block over the whole input, greedy LZ77 with a hash of 3 bytes and short chains.
std::vector<uint8_t> ZlibCompress(const std::vector<uint8_t> &in)
{
    std::vector<uint8_t> out;
    out.reserve(in.size() / 2 + 64);
    out.push_back(0x78); // deflate, 32 K window
    out.push_back(0x5e);
    BitWriter bw { out };
    bw.Bits(1, 1); // BFINAL
    bw.Bits(1, 2); // BTYPE = fixed Huffman

    constexpr int kHashBits = 15, kChain = 16;
    constexpr size_t kWindow = 32768;
    std::vector<int32_t> head(size_t(1) << kHashBits, -1), prev(in.size(), -1);
    auto hash = [&](size_t i) { return ((in[i] << 10) ^ (in[i + 1] << 5) ^ in[i + 2]) & ((1u << kHashBits) - 1); };
    const size_t n = in.size();
    size_t i = 0;
    while (i < n)
    {
        size_t bestLen = 0, bestDist = 0;
        if (i + 3 <= n)
        {
            const uint32_t h = hash(i);
            int32_t cand = head[h];
            for (int c = 0; c < kChain && cand >= 0 && i - static_cast<size_t>(cand) <= kWindow; c++)
            {
                size_t len = 0;
                const size_t maxLen = std::min<size_t>(258, n - i);
                while (len < maxLen && in[static_cast<size_t>(cand) + len] == in[i + len])
                    len++;
                if (len > bestLen)
                {
                    bestLen = len;
                    bestDist = i - static_cast<size_t>(cand);
                    if (len == maxLen)
                        break;
                }
                cand = prev[static_cast<size_t>(cand)];
            }
        }
        const size_t step = bestLen >= 3 ? bestLen : 1;
        if (bestLen >= 3)
        {
            int lc = 28;
            while (kLengthBase[lc] > bestLen)
                lc--;
            PutLiteralOrLength(bw, 257 + static_cast<uint32_t>(lc));
            bw.Bits(static_cast<uint32_t>(bestLen - kLengthBase[lc]), kLengthExtra[lc]);
            int dc = 29;
            while (kDistBase[dc] > bestDist)
                dc--;
            bw.Code(static_cast<uint32_t>(dc), 5);
            bw.Bits(static_cast<uint32_t>(bestDist - kDistBase[dc]), kDistExtra[dc]);
        }
        else
            PutLiteralOrLength(bw, in[i]);
        for (size_t k = i; k < i + step && k + 3 <= n; k++)
        {
            const uint32_t h = hash(k);
            prev[k] = head[h];
            head[h] = static_cast<int32_t>(k);
        }
        i += step;
    }
    PutLiteralOrLength(bw, 256);
    bw.Flush();
    uint32_t a = 1, b = 0; // Adler-32
    for (uint8_t v : in)
    {
        a = (a + v) % 65521u;
        b = (b + a) % 65521u;
    }
    const uint32_t adler = (b << 16) | a;
    for (int s = 24; s >= 0; s -= 8)
        out.push_back(static_cast<uint8_t>(adler >> s));
    return out;
}

uint32_t Crc32(const uint8_t *p, size_t n, uint32_t crc = 0)
{
    static uint32_t table[256];
    static bool ready = false;
    if (!ready)
    {
        for (uint32_t i = 0; i < 256; i++)
        {
            uint32_t c = i;
            for (int k = 0; k < 8; k++)
                c = (c & 1u) ? 0xedb88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        ready = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; i++)
        crc = table[(crc ^ p[i]) & 0xffu] ^ (crc >> 8);
    return ~crc;
}

void PutBE32(std::vector<uint8_t> &v, uint32_t x)
{
    for (int s = 24; s >= 0; s -= 8)
        v.push_back(static_cast<uint8_t>(x >> s));
}

void PutChunk(std::vector<uint8_t> &png, const char type[4], const std::vector<uint8_t> &data)
{
    PutBE32(png, static_cast<uint32_t>(data.size()));
    const size_t start = png.size();
    png.insert(png.end(), type, type + 4);
    png.insert(png.end(), data.begin(), data.end());
    PutBE32(png, Crc32(png.data() + start, png.size() - start));
}

bool WriteFile(const std::filesystem::path &path, const std::vector<uint8_t> &bytes)
{
    FILE *f = std::fopen(path.string().c_str(), "wb");
    if (!f)
        return false;
    const bool ok = std::fwrite(bytes.data(), 1, bytes.size(), f) == bytes.size();
    return std::fclose(f) == 0 && ok;
}

}

// ---- encoders -----------------------------------------------------------------------------------------

std::vector<uint8_t> OutputSaver::EncodePng(uint32_t width, uint32_t height, const uint8_t *rgba)
{
    // scanlines with the Sub filter (type 1): path-traced images are smooth, neighbours predict well
    std::vector<uint8_t> raw;
    raw.reserve((static_cast<size_t>(width) * 4 + 1) * height);
    for (uint32_t y = 0; y < height; y++)
    {
        const uint8_t *row = rgba + static_cast<size_t>(y) * width * 4;
        raw.push_back(1);
        for (uint32_t i = 0; i < width * 4; i++)
            raw.push_back(static_cast<uint8_t>(row[i] - (i >= 4 ? row[i - 4] : 0)));
    }
    std::vector<uint8_t> png = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
    std::vector<uint8_t> ihdr;
    PutBE32(ihdr, width);
    PutBE32(ihdr, height);
    ihdr.insert(ihdr.end(), { 8, 6, 0, 0, 0 }); // 8 bits, RGBA, deflate, adaptive filtering, no interlace
    PutChunk(png, "IHDR", ihdr);
    PutChunk(png, "sRGB", { 0 });
    PutChunk(png, "IDAT", ZlibCompress(raw));
    PutChunk(png, "IEND", {});
    return png;
}

// ---- baseline JPEG (ITU T.81): 4:2:0, quality-scaled Annex K quantisation tables, two passes: the first collects the
// run/size symbol statistics, the second writes the scan with Huffman tables built for exactly this image (Annex K.2) ----

namespace
{

const uint8_t kJpegZigzag[64] = { 0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                  35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };
const uint8_t kLumaQuant[64] = { 16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
                                 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99 };
const uint8_t kChromaQuant[64] = { 17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
                                   99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99 };

struct JpegHuffman
{
    uint8_t bits[17] = {};  // number of codes of each length
    std::vector<uint8_t> vals;
    uint16_t code[256] = {};
    uint8_t size[256] = {};

    // Annex K.2: code lengths from frequencies, limited to 16 bits, no code of all ones
    void Build(const uint32_t *freqIn)
    {
        long freq[257];
        int codesize[257] = {}, others[257];
        for (int i = 0; i < 256; i++)
            freq[i] = freqIn[i];
        freq[256] = 1; // reserves the all-ones code
        for (int i = 0; i < 257; i++)
            others[i] = -1;
        for (;;)
        {
            int c1 = -1, c2 = -1;
            long v = 0x7fffffffL;
            for (int i = 0; i <= 256; i++)
                if (freq[i] && freq[i] <= v) { v = freq[i]; c1 = i; }
            v = 0x7fffffffL;
            for (int i = 0; i <= 256; i++)
                if (freq[i] && freq[i] <= v && i != c1) { v = freq[i]; c2 = i; }
            if (c2 < 0)
                break;
            freq[c1] += freq[c2];
            freq[c2] = 0;
            codesize[c1]++;
            while (others[c1] >= 0) { c1 = others[c1]; codesize[c1]++; }
            others[c1] = c2;
            codesize[c2]++;
            while (others[c2] >= 0) { c2 = others[c2]; codesize[c2]++; }
        }
        int count[33] = {};
        for (int i = 0; i <= 256; i++)
            if (codesize[i])
                count[codesize[i]]++;
        for (int i = 32; i > 16; i--) // shorten codes longer than 16 bits
            while (count[i] > 0)
            {
                int j = i - 2;
                while (count[j] == 0) j--;
                count[i] -= 2;
                count[i - 1]++;
                count[j + 1] += 2;
                count[j]--;
            }
        int i = 16;
        while (count[i] == 0) i--;
        count[i]--; // remove the reserved symbol
        for (int k = 1; k <= 16; k++)
            bits[k] = static_cast<uint8_t>(count[k]);
        vals.clear();
        for (int len = 1; len <= 32; len++)
            for (int sym = 0; sym < 256; sym++)
                if (codesize[sym] == len)
                    vals.push_back(static_cast<uint8_t>(sym));
        uint16_t next = 0;
        size_t k = 0;
        for (int len = 1; len <= 16; len++)
        {
            for (int n = 0; n < bits[len]; n++, k++)
            {
                code[vals[k]] = next++;
                size[vals[k]] = static_cast<uint8_t>(len);
            }
            next = static_cast<uint16_t>(next << 1);
        }
    }
};

void ForwardDct(float *b) // separable 8x8 DCT-II, T.81 A.3.3
{
    static float c[8][8];
    static bool ready = false;
    if (!ready)
    {
        for (int u = 0; u < 8; u++)
            for (int x = 0; x < 8; x++)
                c[u][x] = (u == 0 ? 0.35355339f : 0.5f) * std::cos((2 * x + 1) * u * 3.14159265358979f / 16.0f);
        ready = true;
    }
    float tmp[64];
    for (int y = 0; y < 8; y++)
        for (int u = 0; u < 8; u++)
        {
            float s = 0;
            for (int x = 0; x < 8; x++)
                s += c[u][x] * b[y * 8 + x];
            tmp[y * 8 + u] = s;
        }
    for (int u = 0; u < 8; u++)
        for (int v = 0; v < 8; v++)
        {
            float s = 0;
            for (int y = 0; y < 8; y++)
                s += c[v][y] * tmp[y * 8 + u];
            b[v * 8 + u] = s;
        }
}

int BitLength(int v)
{
    int n = 0;
    for (v = v < 0 ? -v : v; v; v >>= 1)
        n++;
    return n;
}

}

std::vector<uint8_t> OutputSaver::EncodeJpg(uint32_t width, uint32_t height, const uint8_t *rgba, int quality)
{
    quality = quality <= 0 ? 90 : std::min(quality, 100);
    const int scale = quality < 50 ? 5000 / quality : 200 - 2 * quality;
    uint8_t qt[2][64];
    for (int i = 0; i < 64; i++)
    {
        qt[0][i] = static_cast<uint8_t>(std::clamp((kLumaQuant[i] * scale + 50) / 100, 1, 255));
        qt[1][i] = static_cast<uint8_t>(std::clamp((kChromaQuant[i] * scale + 50) / 100, 1, 255));
    }
    const uint32_t mcusX = (width + 15) / 16, mcusY = (height + 15) / 16;
    // planes padded to whole MCUs by edge replication: Y at full, Cb / Cr at half resolution (2x2 box)
    const uint32_t pw = mcusX * 16, ph = mcusY * 16;
    std::vector<float> Y(static_cast<size_t>(pw) * ph), Cb(static_cast<size_t>(pw / 2) * (ph / 2)), Cr(Cb.size());
    for (uint32_t y = 0; y < ph; y++)
        for (uint32_t x = 0; x < pw; x++)
        {
            const uint8_t *p = rgba + (static_cast<size_t>(std::min(y, height - 1)) * width + std::min(x, width - 1)) * 4;
            const float r = p[0], g = p[1], b = p[2];
            Y[static_cast<size_t>(y) * pw + x] = 0.299f * r + 0.587f * g + 0.114f * b - 128.0f;
            const size_t c = static_cast<size_t>(y / 2) * (pw / 2) + x / 2;
            Cb[c] += 0.25f * (-0.168736f * r - 0.331264f * g + 0.5f * b);
            Cr[c] += 0.25f * (0.5f * r - 0.418688f * g - 0.081312f * b);
        }

    // quantised coefficient blocks in scan order: per MCU four Y blocks, Cb, Cr
    struct Block { int16_t q[64]; int comp; };
    std::vector<Block> blocks;
    blocks.reserve(static_cast<size_t>(mcusX) * mcusY * 6);
    auto encodeBlock = [&](const std::vector<float> &plane, uint32_t stride, uint32_t bx, uint32_t by, int comp) {
        float b[64];
        for (int y = 0; y < 8; y++)
            for (int x = 0; x < 8; x++)
                b[y * 8 + x] = plane[static_cast<size_t>(by * 8 + y) * stride + bx * 8 + x];
        ForwardDct(b);
        Block out;
        out.comp = comp;
        const uint8_t *q = qt[comp ? 1 : 0];
        for (int i = 0; i < 64; i++)
        {
            const int z = kJpegZigzag[i];
            out.q[i] = static_cast<int16_t>(std::lround(b[z] / static_cast<float>(q[z])));
        }
        blocks.push_back(out);
    };
    for (uint32_t my = 0; my < mcusY; my++)
        for (uint32_t mx = 0; mx < mcusX; mx++)
        {
            for (uint32_t k = 0; k < 4; k++)
                encodeBlock(Y, pw, mx * 2 + (k & 1), my * 2 + (k >> 1), 0);
            encodeBlock(Cb, pw / 2, mx, my, 1);
            encodeBlock(Cr, pw / 2, mx, my, 2);
        }

    // pass 1: symbol statistics; pass 2: emit
    uint32_t freq[4][256] = {}; // DC luma, AC luma, DC chroma, AC chroma
    JpegHuffman huff[4];
    std::vector<uint8_t> scan;
    uint32_t acc = 0;
    int nbits = 0;
    auto put = [&](uint32_t code, int len) {
        acc = (acc << len) | (code & ((1u << len) - 1u));
        nbits += len;
        while (nbits >= 8)
        {
            const uint8_t byte = static_cast<uint8_t>(acc >> (nbits - 8));
            scan.push_back(byte);
            if (byte == 0xff)
                scan.push_back(0);
            nbits -= 8;
        }
    };
    for (int pass = 0; pass < 2; pass++)
    {
        int pred[3] = { 0, 0, 0 };
        for (const Block &blk : blocks)
        {
            const int t = blk.comp ? 2 : 0;
            auto symbol = [&](int table, int sym, int value, int len) {
                if (pass == 0)
                    freq[table][sym]++;
                else
                {
                    put(huff[table].code[sym], huff[table].size[sym]);
                    if (len)
                        put(static_cast<uint32_t>(value < 0 ? value + (1 << len) - 1 : value), len);
                }
            };
            const int diff = blk.q[0] - pred[blk.comp];
            pred[blk.comp] = blk.q[0];
            const int dl = BitLength(diff);
            symbol(t, dl, diff, dl);
            int run = 0, last = 63;
            while (last > 0 && blk.q[last] == 0)
                last--;
            for (int i = 1; i <= last; i++)
            {
                if (blk.q[i] == 0) { run++; continue; }
                while (run > 15) { symbol(t + 1, 0xf0, 0, 0); run -= 16; }
                const int len = BitLength(blk.q[i]);
                symbol(t + 1, (run << 4) | len, blk.q[i], len);
                run = 0;
            }
            if (last < 63)
                symbol(t + 1, 0x00, 0, 0);
        }
        if (pass == 0)
            for (int t = 0; t < 4; t++)
                huff[t].Build(freq[t]);
    }
    if (nbits)
        put(0x7f, 8 - nbits); // pad the last byte with ones

    std::vector<uint8_t> out = { 0xff, 0xd8 };
    auto segment = [&](uint8_t marker, const std::vector<uint8_t> &body) {
        out.push_back(0xff);
        out.push_back(marker);
        out.push_back(static_cast<uint8_t>((body.size() + 2) >> 8));
        out.push_back(static_cast<uint8_t>((body.size() + 2) & 0xff));
        out.insert(out.end(), body.begin(), body.end());
    };
    segment(0xe0, { 'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0 });
    for (int t = 0; t < 2; t++)
    {
        std::vector<uint8_t> dqt = { static_cast<uint8_t>(t) };
        for (int i = 0; i < 64; i++)
            dqt.push_back(qt[t][kJpegZigzag[i]]);
        segment(0xdb, dqt);
    }
    segment(0xc0, { 8, static_cast<uint8_t>(height >> 8), static_cast<uint8_t>(height & 0xff), static_cast<uint8_t>(width >> 8), static_cast<uint8_t>(width & 0xff), 3,
                    1, 0x22, 0, 2, 0x11, 1, 3, 0x11, 1 });
    for (int t = 0; t < 4; t++)
    {
        std::vector<uint8_t> dht = { static_cast<uint8_t>(((t & 1) << 4) | (t >> 1)) }; // class (DC / AC) and destination (luma / chroma)
        for (int k = 1; k <= 16; k++)
            dht.push_back(huff[t].bits[k]);
        dht.insert(dht.end(), huff[t].vals.begin(), huff[t].vals.end());
        segment(0xc4, dht);
    }
    segment(0xda, { 3, 1, 0x00, 2, 0x11, 3, 0x11, 0, 63, 0 });
    out.insert(out.end(), scan.begin(), scan.end());
    out.push_back(0xff);
    out.push_back(0xd9);
    return out;
}

std::vector<uint8_t> OutputSaver::EncodeTga(uint32_t width, uint32_t height, const uint8_t *rgba)
{
    std::vector<uint8_t> tga(18, 0);
    tga[2] = 2; // uncompressed true colour
    tga[12] = static_cast<uint8_t>(width & 0xff); tga[13] = static_cast<uint8_t>(width >> 8);
    tga[14] = static_cast<uint8_t>(height & 0xff); tga[15] = static_cast<uint8_t>(height >> 8);
    tga[16] = 32;
    tga[17] = 0x28; // 8 alpha bits, top-left origin
    tga.reserve(18 + static_cast<size_t>(width) * height * 4);
    for (size_t i = 0; i < static_cast<size_t>(width) * height; i++)
    {
        const uint8_t *p = rgba + i * 4;
        tga.insert(tga.end(), { p[2], p[1], p[0], p[3] });
    }
    return tga;
}

std::vector<uint8_t> OutputSaver::EncodeHdr(uint32_t width, uint32_t height, const float *rgba)
{
    const std::string header = "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y " + std::to_string(height) + " +X " + std::to_string(width) + "\n";
    std::vector<uint8_t> hdr(header.begin(), header.end());
    hdr.reserve(hdr.size() + static_cast<size_t>(width) * height * 4);
    for (size_t i = 0; i < static_cast<size_t>(width) * height; i++) // flat (non run-length) RGBE scanlines
    {
        const float r = rgba[i * 4], g = rgba[i * 4 + 1], b = rgba[i * 4 + 2];
        const float m = std::max(r, std::max(g, b));
        if (!(m > 1e-32f))
        {
            hdr.insert(hdr.end(), { 0, 0, 0, 0 });
            continue;
        }
        int e;
        const float scale = std::frexp(m, &e) * 256.0f / m;
        hdr.insert(hdr.end(), { static_cast<uint8_t>(r * scale), static_cast<uint8_t>(g * scale), static_cast<uint8_t>(b * scale),
                                static_cast<uint8_t>(e + 128) });
    }
    return hdr;
}

// ---- OutputSaver ------------------------------------------------------------------------------------------

OutputSaver::OutputSaver()
{
    m_HasFFmpeg = std::system("ffmpeg -version > /dev/null 2>&1") == 0; // OutputSaver.cpp:32-46
}

OutputSaver::~OutputSaver()
{
    EndOutput();
}

uint32_t OutputSaver::SelectImageFormat(OutputFormat format)
{
    return format == OutputFormat::Hdr ? PTX_OUTPUT_RGBA32F : PTX_OUTPUT_RGBA8_SRGB;
}

void OutputSaver::RegisterOutput(const OutputInfo &info)
{
    EndOutput();
    m_Info = info;
    m_Registered = true;
    if (info.Format == OutputFormat::Mp4 && m_HasFFmpeg) // the command line of OutputSaver.cpp:88-97
    {
        const std::string cmd = "ffmpeg -r " + std::to_string(info.Framerate) + " -f rawvideo -pix_fmt rgba -s " + std::to_string(info.Extent.width) +
                                "x" + std::to_string(info.Extent.height) +
                                " -i - -y -an -vcodec libx264 -preset veryslow -crf 17 -pix_fmt yuv420p -threads 0 \"" + info.Path.string() +
                                "\" > /dev/null 2>&1";
        m_FFmpegPipe = popen(cmd.c_str(), "w");
    }
}

bool OutputSaver::SubmitFrame(std::span<const std::byte> data)
{
    if (!m_Registered)
        return false;
    if (m_Info.Format == OutputFormat::Mp4 && !m_FFmpegPipe)
        return false;
    return WriteImage(m_Info, data, m_FFmpegPipe);
}

void OutputSaver::EndOutput()
{
    if (m_FFmpegPipe)
    {
        pclose(m_FFmpegPipe);
        m_FFmpegPipe = nullptr;
    }
    m_Registered = false;
}

void OutputSaver::CancelOutput()
{
    const bool had = m_Registered;
    EndOutput();
    if (had)
    {
        std::error_code ec;
        std::filesystem::remove(m_Info.Path, ec); // OutputSaver.cpp:223
    }
}

bool OutputSaver::WriteImage(const OutputInfo &info, std::span<const std::byte> data, FILE *videoPipe)
{
    const uint32_t w = info.Extent.width, h = info.Extent.height;
    const size_t texels = static_cast<size_t>(w) * h;
    switch (info.Format)
    {
    case OutputFormat::Png:
        return data.size() == texels * 4 && WriteFile(info.Path, EncodePng(w, h, reinterpret_cast<const uint8_t *>(data.data())));
    case OutputFormat::Tga:
        return data.size() == texels * 4 && WriteFile(info.Path, EncodeTga(w, h, reinterpret_cast<const uint8_t *>(data.data())));
    case OutputFormat::Hdr:
        return data.size() == texels * 16 && WriteFile(info.Path, EncodeHdr(w, h, reinterpret_cast<const float *>(data.data())));
    case OutputFormat::Mp4:
        return videoPipe && data.size() == texels * 4 && std::fwrite(data.data(), data.size(), 1, videoPipe) == 1;
    case OutputFormat::Jpg:
        return data.size() == texels * 4 && w < 65536 && h < 65536 && WriteFile(info.Path, EncodeJpg(w, h, reinterpret_cast<const uint8_t *>(data.data())));
    default:
        return false;
    }
}

// ---- checkpoint ----------------------------------------------------------------------------------------------

bool SaveCheckpoint(const std::filesystem::path &path, uint32_t width, uint32_t height, uint32_t totalSamples, const float *rgba)
{
    FILE *f = std::fopen(path.string().c_str(), "wb");
    if (!f)
        return false;
    uint32_t header[8] = { 0, 0, width, height, totalSamples, 0, 0, 0 };
    std::memcpy(header, "PTXACC1", 8);
    const size_t n = static_cast<size_t>(width) * height * 4;
    const bool ok = std::fwrite(header, sizeof(header), 1, f) == 1 && std::fwrite(rgba, sizeof(float), n, f) == n;
    return std::fclose(f) == 0 && ok;
}

bool LoadCheckpoint(const std::filesystem::path &path, uint32_t &width, uint32_t &height, uint32_t &totalSamples, std::vector<float> &rgba)
{
    FILE *f = std::fopen(path.string().c_str(), "rb");
    if (!f)
        return false;
    uint32_t header[8];
    bool ok = std::fread(header, sizeof(header), 1, f) == 1 && std::memcmp(header, "PTXACC1", 8) == 0;
    if (ok)
    {
        width = header[2];
        height = header[3];
        totalSamples = header[4];
        const size_t n = static_cast<size_t>(width) * height * 4;
        ok = width && height && n / 4 / width == height;
        if (ok)
        {
            rgba.resize(n);
            ok = std::fread(rgba.data(), sizeof(float), n, f) == n;
        }
    }
    std::fclose(f);
    return ok;
}

}
