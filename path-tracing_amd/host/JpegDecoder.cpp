// JpegDecoder.cpp -- TextureImporter::DecodeJpeg: baseline / extended sequential and progressive Huffman JPEG (ITU T.81,
// SOF0 / SOF1 / SOF2), 8-bit samples, 1 or 3 components, arbitrary sampling factors, restart intervals.  Chroma subsampled 2:1 in either
// direction is upsampled with the triangle filter stb_image and libjpeg use (3/4 near + 1/4 far per axis), other factors
// by replication; colour conversion is the JFIF YCbCr matrix.  Lossless and arithmetic-coded files are rejected
// (the importer then falls back to the slot's default texture, as the reference does for any load failure).
#include <algorithm>
#include <cmath>
#include <cstring>

#include "TextureImporter.h"

namespace PathTracing
{

namespace
{

struct HuffTable
{
    uint8_t bits[17] = {};
    uint8_t vals[256] = {};
    int mincode[17], maxcode[18], valptr[17];
    bool present = false;
    void Build()
    {
        int code = 0, k = 0;
        for (int len = 1; len <= 16; len++)
        {
            valptr[len] = k;
            mincode[len] = code;
            code += bits[len];
            k += bits[len];
            maxcode[len] = bits[len] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        present = true;
    }
};

struct Component
{
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int pred = 0;
    int bw = 0, bh = 0; // size of the component plane in blocks (padded to whole MCUs)
    std::vector<uint8_t> plane;
    std::vector<int16_t> coef; // progressive: 64 coefficients per block of the padded plane, in natural order
};

struct Reader
{
    const uint8_t *p, *end;
    uint32_t acc = 0;
    int n = 0;
    bool hitMarker = false;
    int Bit()
    {
        if (n == 0)
        {
            uint8_t b = 0;
            if (p < end && !hitMarker)
            {
                b = *p++;
                if (b == 0xff)
                {
                    const uint8_t m = p < end ? *p : 0;
                    if (m == 0)
                        p++; // stuffed zero
                    else
                    {
                        hitMarker = true; // feed zeros from here on; the marker is handled by the caller
                        p--;
                        b = 0;
                    }
                }
            }
            acc = b;
            n = 8;
        }
        n--;
        return (acc >> n) & 1;
    }
    int Bits(int count)
    {
        int v = 0;
        while (count--)
            v = (v << 1) | Bit();
        return v;
    }
    void Reset()
    {
        n = 0;
        hitMarker = false;
    }
};

int DecodeSymbol(Reader &r, const HuffTable &t)
{
    int code = 0;
    for (int len = 1; len <= 16; len++)
    {
        code = (code << 1) | r.Bit();
        if (t.maxcode[len] >= 0 && code <= t.maxcode[len] && code >= t.mincode[len])
            return t.vals[t.valptr[len] + code - t.mincode[len]];
    }
    throw error("JPEG: bad Huffman code");
}

inline int Extend(int v, int bits)
{
    return bits && v < (1 << (bits - 1)) ? v - (1 << bits) + 1 : v;
}

const uint8_t kZigzag[64] = { 0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                              35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };

// separable 8x8 inverse DCT in float (T.81 A.3.3), + 128 level shift, clamped to 8 bits
void Idct(const float *in, uint8_t *out, int stride)
{
    static float c[8][8];
    static bool ready = false;
    if (!ready)
    {
        for (int x = 0; x < 8; x++)
            for (int u = 0; u < 8; u++)
                c[x][u] = (u == 0 ? 0.35355339f : 0.5f) * std::cos((2 * x + 1) * u * 3.14159265358979f / 16.0f);
        ready = true;
    }
    float tmp[64];
    for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++)
        {
            float s = 0;
            for (int u = 0; u < 8; u++)
                s += c[x][u] * in[y * 8 + u];
            tmp[y * 8 + x] = s;
        }
    for (int x = 0; x < 8; x++)
        for (int y = 0; y < 8; y++)
        {
            float s = 0;
            for (int v = 0; v < 8; v++)
                s += c[y][v] * tmp[v * 8 + x];
            const int q = static_cast<int>(std::floor(s + 128.5f));
            out[y * stride + x] = static_cast<uint8_t>(q < 0 ? 0 : (q > 255 ? 255 : q));
        }
}

}

DecodedImage TextureImporter::DecodeJpeg(std::span<const uint8_t> f)
{
    if (f.size() < 4 || f[0] != 0xff || f[1] != 0xd8)
        throw error("Not a JPEG file");
    uint16_t quant[4][64] = {};
    HuffTable dc[4], ac[4];
    Component comp[3];
    int ncomp = 0, width = 0, height = 0, restartInterval = 0, hmax = 1, vmax = 1;
    bool haveFrame = false, adobe = false, progressive = false;
    int adobeTransform = -1;
    size_t pos = 2;
    auto be16 = [&](size_t o) { return (f[o] << 8) | f[o + 1]; };
    for (;;)
    {
        while (pos < f.size() && f[pos] != 0xff) pos++;
        while (pos < f.size() && f[pos] == 0xff) pos++;
        if (pos >= f.size())
            throw error("JPEG: no image data");
        const int marker = f[pos++];
        if (marker == 0xd9)
            throw error("JPEG: no scan");
        if (pos + 2 > f.size())
            throw error("JPEG: truncated segment");
        const size_t len = static_cast<size_t>(be16(pos));
        if (len < 2 || pos + len > f.size())
            throw error("JPEG: bad segment length");
        const size_t seg = pos + 2, segEnd = pos + len;
        if (marker == 0xdb) // DQT
        {
            size_t q = seg;
            while (q < segEnd)
            {
                const int pq = f[q] >> 4, tq = f[q] & 15;
                q++;
                if (tq > 3 || q + (pq ? 128 : 64) > segEnd) throw error("JPEG: bad DQT");
                for (int i = 0; i < 64; i++)
                {
                    quant[tq][kZigzag[i]] = static_cast<uint16_t>(pq ? be16(q) : f[q]);
                    q += pq ? 2 : 1;
                }
            }
        }
        else if (marker == 0xc4) // DHT
        {
            size_t q = seg;
            while (q + 17 <= segEnd)
            {
                const int tc = f[q] >> 4, th = f[q] & 15;
                if (tc > 1 || th > 3) throw error("JPEG: bad DHT");
                HuffTable &t = tc ? ac[th] : dc[th];
                int total = 0;
                for (int i = 1; i <= 16; i++) { t.bits[i] = f[q + i]; total += t.bits[i]; }
                q += 17;
                if (total > 256 || q + total > segEnd) throw error("JPEG: bad DHT");
                std::memcpy(t.vals, &f[q], static_cast<size_t>(total));
                q += static_cast<size_t>(total);
                t.Build();
            }
        }
        else if (marker == 0xc0 || marker == 0xc1 || marker == 0xc2) // SOF0 / SOF1 / SOF2 (progressive)
        {
            progressive = marker == 0xc2;
            if (len < 8 || f[seg] != 8) throw error("JPEG: only 8-bit samples are supported");
            if (haveFrame) throw error("JPEG: more than one frame header");
            height = be16(seg + 1);
            width = be16(seg + 3);
            ncomp = f[seg + 5];
            if (!(ncomp == 1 || ncomp == 3) || !width || !height || seg + 6 + static_cast<size_t>(ncomp) * 3 > segEnd)
                throw error("JPEG: unsupported component count");
            if (static_cast<uint64_t>(width) * static_cast<uint64_t>(height) > 16384ull * 16384ull)
                throw error("JPEG: image too large");
            for (int i = 0; i < ncomp; i++)
            {
                comp[i].id = f[seg + 6 + i * 3];
                comp[i].h = f[seg + 7 + i * 3] >> 4;
                comp[i].v = f[seg + 7 + i * 3] & 15;
                comp[i].tq = f[seg + 8 + i * 3];
                if (comp[i].h < 1 || comp[i].h > 4 || comp[i].v < 1 || comp[i].v > 4 || comp[i].tq > 3) throw error("JPEG: bad SOF");
                hmax = std::max(hmax, comp[i].h);
                vmax = std::max(vmax, comp[i].v);
            }
            haveFrame = true;
        }
        else if (marker >= 0xc3 && marker <= 0xcf && marker != 0xc8 && marker != 0xcc)
            throw error("JPEG: lossless / arithmetic files are not supported");
        else if (marker == 0xdd)
        {
            if (len < 4) throw error("JPEG: bad DRI");
            restartInterval = be16(seg);
        }
        else if (marker == 0xee && len >= 14 && !std::memcmp(&f[seg], "Adobe", 5))
        {
            adobe = true;
            adobeTransform = f[seg + 11];
        }
        else if (marker == 0xda) // SOS
        {
            if (!haveFrame) throw error("JPEG: scan before frame");
            if (seg >= segEnd) throw error("JPEG: bad SOS");
            const int ns = f[seg];
            if (progressive)
                break; // the scans of a progressive file are read below, starting again from this marker
            if (ns != ncomp) throw error("JPEG: multi-scan files are not supported");
            if (seg + 1 + 2 * static_cast<size_t>(ns) + 3 > segEnd) throw error("JPEG: bad SOS");
            for (int i = 0; i < ns; i++)
            {
                const int cs = f[seg + 1 + i * 2], tdta = f[seg + 2 + i * 2];
                int k = 0;
                while (k < ncomp && comp[k].id != cs) k++;
                if (k == ncomp) throw error("JPEG: bad SOS");
                comp[k].td = tdta >> 4;
                comp[k].ta = tdta & 15;
                if (comp[k].td > 3 || comp[k].ta > 3 || !dc[comp[k].td].present || !ac[comp[k].ta].present) throw error("JPEG: missing Huffman table");
            }
            pos = segEnd;
            break;
        }
        pos = segEnd;
    }

    const int mcuW = 8 * hmax, mcuH = 8 * vmax;
    const int mcusX = (width + mcuW - 1) / mcuW, mcusY = (height + mcuH - 1) / mcuH;
    for (int i = 0; i < ncomp; i++)
    {
        if (ncomp == 1) { comp[i].h = comp[i].v = 1; }
        comp[i].bw = mcusX * comp[i].h;
        comp[i].bh = mcusY * comp[i].v;
        comp[i].plane.assign(static_cast<size_t>(comp[i].bw) * 8 * comp[i].bh * 8, 0);
    }
    if (ncomp == 1) { hmax = vmax = 1; }
    const int mcusX1 = ncomp == 1 ? (width + 7) / 8 : mcusX, mcusY1 = ncomp == 1 ? (height + 7) / 8 : mcusY;
    if (ncomp == 1)
    {
        comp[0].bw = mcusX1;
        comp[0].bh = mcusY1;
        comp[0].plane.assign(static_cast<size_t>(mcusX1) * 8 * mcusY1 * 8, 0);
    }

    if (progressive)
    {
        // ---- T.81 Annex G: several scans, each a band of coefficients (Ss..Se) at some precision (Ah, Al) of one component,
        // or the DC terms of all of them; coefficients accumulate in `coef` and are transformed once at the end
        for (int i = 0; i < ncomp; i++)
            comp[i].coef.assign(static_cast<size_t>(comp[i].bw) * comp[i].bh * 64, 0);
        size_t at = pos - 1; // `pos` is just past the SOS marker byte: step back onto it
        while (at > 0 && f[at] != 0xda) at--;
        at--; // the 0xff in front of it
        int scans = 0;
        for (;;)
        {
            while (at < f.size() && f[at] != 0xff) at++;
            while (at < f.size() && f[at] == 0xff) at++;
            if (at >= f.size())
                break; // no EOI: use what has arrived
            const int marker = f[at++];
            if (marker == 0xd9)
                break;
            if (marker >= 0xd0 && marker <= 0xd7)
                continue;
            if (at + 2 > f.size()) throw error("JPEG: truncated segment");
            const size_t len = static_cast<size_t>(be16(at));
            if (len < 2 || at + len > f.size()) throw error("JPEG: bad segment length");
            const size_t seg = at + 2, segEnd = at + len;
            if (marker == 0xc4) // DHT between scans
            {
                size_t q = seg;
                while (q + 17 <= segEnd)
                {
                    const int tc = f[q] >> 4, th = f[q] & 15;
                    if (tc > 1 || th > 3) throw error("JPEG: bad DHT");
                    HuffTable &t = tc ? ac[th] : dc[th];
                    int total = 0;
                    for (int i = 1; i <= 16; i++) { t.bits[i] = f[q + i]; total += t.bits[i]; }
                    q += 17;
                    if (total > 256 || q + total > segEnd) throw error("JPEG: bad DHT");
                    std::memcpy(t.vals, &f[q], static_cast<size_t>(total));
                    q += static_cast<size_t>(total);
                    t.Build();
                }
            }
            else if (marker == 0xdd)
            {
                if (len < 4) throw error("JPEG: bad DRI");
                restartInterval = be16(seg);
            }
            else if (marker == 0xdb)
                throw error("JPEG: quantisation tables redefined between scans are not supported");
            if (marker != 0xda)
            {
                at = segEnd;
                continue;
            }
            // ---- one scan
            if (seg >= segEnd) throw error("JPEG: bad SOS");
            const int ns = f[seg];
            if (ns < 1 || ns > ncomp || seg + 1 + 2 * static_cast<size_t>(ns) + 3 > segEnd) throw error("JPEG: bad SOS");
            int which[3] = { 0, 0, 0 };
            for (int i = 0; i < ns; i++)
            {
                const int cs = f[seg + 1 + i * 2], tdta = f[seg + 2 + i * 2];
                int k = 0;
                while (k < ncomp && comp[k].id != cs) k++;
                if (k == ncomp) throw error("JPEG: bad SOS");
                which[i] = k;
                comp[k].td = tdta >> 4;
                comp[k].ta = tdta & 15;
                if (comp[k].td > 3 || comp[k].ta > 3) throw error("JPEG: bad SOS");
            }
            const int Ss = f[seg + 1 + 2 * ns], Se = f[seg + 2 + 2 * ns], Ah = f[seg + 3 + 2 * ns] >> 4, Al = f[seg + 3 + 2 * ns] & 15;
            if (Ss > Se || Se > 63 || (Ss == 0 && Se != 0) || (Ss > 0 && ns != 1) || Al > 13) throw error("JPEG: bad progressive scan parameters");
            for (int i = 0; i < ns; i++)
                if ((Ss == 0 && Ah == 0 && !dc[comp[which[i]].td].present) || (Ss > 0 && !ac[comp[which[i]].ta].present))
                    throw error("JPEG: missing Huffman table");
            if (++scans > 1000) throw error("JPEG: too many scans");
            Reader r { f.data() + segEnd, f.data() + f.size() };
            int eobrun = 0, restartCount = 0;
            for (int i = 0; i < ncomp; i++) comp[i].pred = 0;
            const int p1 = 1 << Al, m1 = -(1 << Al);
            auto block = [&](Component &c, int bx, int by) {
                int16_t *b = &c.coef[(static_cast<size_t>(by) * c.bw + bx) * 64];
                if (Ss == 0)
                {
                    if (Ah == 0) // DC first
                    {
                        const int t = DecodeSymbol(r, dc[c.td]);
                        if (t > 11) throw error("JPEG: bad DC category");
                        c.pred += Extend(r.Bits(t), t);
                        b[0] = static_cast<int16_t>(c.pred * p1);
                    }
                    else if (r.Bit()) // DC refinement
                        b[0] = static_cast<int16_t>(b[0] | p1);
                    return;
                }
                if (Ah == 0) // AC first
                {
                    if (eobrun > 0)
                    {
                        eobrun--;
                        return;
                    }
                    for (int k = Ss; k <= Se;)
                    {
                        const int rs = DecodeSymbol(r, ac[c.ta]), run = rs >> 4, size = rs & 15;
                        if (size == 0)
                        {
                            if (run < 15)
                            {
                                eobrun = (1 << run) - 1;
                                if (run) eobrun += r.Bits(run);
                                break;
                            }
                            k += 16;
                            continue;
                        }
                        k += run;
                        if (k > Se) throw error("JPEG: bad AC run");
                        b[kZigzag[k]] = static_cast<int16_t>(Extend(r.Bits(size), size) * p1);
                        k++;
                    }
                    return;
                }
                // AC refinement (G.1.2.3)
                int k = Ss;
                auto correct = [&](int16_t &v) {
                    if (r.Bit() && (v & p1) == 0)
                        v = static_cast<int16_t>(v + (v >= 0 ? p1 : m1));
                };
                if (eobrun == 0)
                {
                    for (; k <= Se; k++)
                    {
                        const int rs = DecodeSymbol(r, ac[c.ta]);
                        int run = rs >> 4;
                        int value = 0;
                        if (rs & 15)
                            value = r.Bit() ? p1 : m1; // a newly non-zero coefficient: always magnitude 1 at this precision
                        else if (run != 15)
                        {
                            eobrun = 1 << run;
                            if (run) eobrun += r.Bits(run);
                            break;
                        }
                        while (k <= Se)
                        {
                            int16_t &v = b[kZigzag[k]];
                            if (v != 0)
                                correct(v);
                            else if (--run < 0)
                                break;
                            k++;
                        }
                        if (value && k <= Se)
                            b[kZigzag[k]] = static_cast<int16_t>(value);
                    }
                }
                if (eobrun > 0)
                {
                    for (; k <= Se; k++)
                        if (b[kZigzag[k]] != 0)
                            correct(b[kZigzag[k]]);
                    eobrun--;
                }
            };
            auto restartIfDue = [&]() {
                if (restartInterval && restartCount == restartInterval)
                {
                    r.Reset();
                    while (r.p + 1 < r.end && !(r.p[0] == 0xff && r.p[1] >= 0xd0 && r.p[1] <= 0xd7)) r.p++;
                    if (r.p + 1 < r.end) r.p += 2;
                    for (int i = 0; i < ncomp; i++) comp[i].pred = 0;
                    eobrun = 0;
                    restartCount = 0;
                }
                restartCount++;
            };
            if (ns > 1) // interleaved (DC) scan: MCU order
            {
                for (int my = 0; my < mcusY1; my++)
                    for (int mx = 0; mx < mcusX1; mx++)
                    {
                        restartIfDue();
                        for (int i = 0; i < ns; i++)
                        {
                            Component &c = comp[which[i]];
                            for (int by = 0; by < c.v; by++)
                                for (int bx = 0; bx < c.h; bx++)
                                    block(c, mx * c.h + bx, my * c.v + by);
                        }
                    }
            }
            else // one component: its blocks row by row, only those that carry image samples
            {
                Component &c = comp[which[0]];
                const int cw = (width * c.h + hmax - 1) / hmax, chh = (height * c.v + vmax - 1) / vmax;
                const int blocksX = (cw + 7) / 8, blocksY = (chh + 7) / 8;
                for (int by = 0; by < blocksY; by++)
                    for (int bx = 0; bx < blocksX; bx++)
                    {
                        restartIfDue();
                        block(c, bx, by);
                    }
            }
            at = static_cast<size_t>(r.p - f.data()); // on or before the next marker
            if (at < segEnd) at = segEnd;
        }
        if (!scans) throw error("JPEG: no scan");
        for (int i = 0; i < ncomp; i++)
            for (int by = 0; by < comp[i].bh; by++)
                for (int bx = 0; bx < comp[i].bw; bx++)
                {
                    const int16_t *b = &comp[i].coef[(static_cast<size_t>(by) * comp[i].bw + bx) * 64];
                    float deq[64];
                    for (int z = 0; z < 64; z++)
                        deq[z] = static_cast<float>(b[z] * quant[comp[i].tq][z]);
                    const int stride = comp[i].bw * 8;
                    Idct(deq, &comp[i].plane[static_cast<size_t>(by) * 8 * stride + bx * 8], stride);
                }
    }
    Reader r { f.data() + pos, f.data() + f.size() };
    int restartCount = 0;
    for (int my = 0; !progressive && my < mcusY1; my++)
        for (int mx = 0; mx < mcusX1; mx++)
        {
            if (restartInterval && restartCount == restartInterval)
            {
                // expect RSTn: skip to the marker, reset predictions
                r.Reset();
                while (r.p + 1 < r.end && !(r.p[0] == 0xff && r.p[1] >= 0xd0 && r.p[1] <= 0xd7)) r.p++;
                if (r.p + 1 < r.end) r.p += 2;
                for (int i = 0; i < ncomp; i++) comp[i].pred = 0;
                restartCount = 0;
            }
            for (int i = 0; i < ncomp; i++)
                for (int by = 0; by < comp[i].v; by++)
                    for (int bx = 0; bx < comp[i].h; bx++)
                    {
                        float block[64] = {};
                        const int t = DecodeSymbol(r, dc[comp[i].td]);
                        if (t > 11) throw error("JPEG: bad DC category");
                        comp[i].pred += Extend(r.Bits(t), t);
                        block[0] = static_cast<float>(comp[i].pred * quant[comp[i].tq][0]);
                        for (int k = 1; k < 64;)
                        {
                            const int rs = DecodeSymbol(r, ac[comp[i].ta]), run = rs >> 4, size = rs & 15;
                            if (size == 0)
                            {
                                if (run != 15) break; // EOB
                                k += 16;
                                continue;
                            }
                            k += run;
                            if (k > 63) throw error("JPEG: bad AC run");
                            const int z = kZigzag[k];
                            block[z] = static_cast<float>(Extend(r.Bits(size), size) * quant[comp[i].tq][z]);
                            k++;
                        }
                        const int px = (mx * comp[i].h + bx) * 8, py = (my * comp[i].v + by) * 8, stride = comp[i].bw * 8;
                        Idct(block, &comp[i].plane[static_cast<size_t>(py) * stride + px], stride);
                    }
            restartCount++;
        }

    DecodedImage img;
    img.Width = static_cast<uint32_t>(width);
    img.Height = static_cast<uint32_t>(height);
    img.Channels = static_cast<uint32_t>(ncomp);
    img.Pixels.resize(static_cast<size_t>(width) * height * 4);
    // full-resolution planes: 2:1 triangle-filter upsampling per axis where the component is subsampled by two
    std::vector<std::vector<uint8_t>> full(static_cast<size_t>(ncomp));
    for (int i = 0; i < ncomp; i++)
    {
        const int sw = comp[i].bw * 8, sh = comp[i].bh * 8; // stored plane
        const int cw = (width * comp[i].h + hmax - 1) / hmax, chh = (height * comp[i].v + vmax - 1) / vmax; // meaningful part
        const bool h2 = hmax == 2 * comp[i].h, v2 = vmax == 2 * comp[i].v;
        std::vector<uint8_t> &dst = full[static_cast<size_t>(i)];
        dst.resize(static_cast<size_t>(width) * height);
        if ((hmax == comp[i].h || h2) && (vmax == comp[i].v || v2) && (h2 || v2))
        {
            std::vector<int> row(static_cast<size_t>(cw));
            for (int y = 0; y < height; y++)
            {
                // vertical: 3 * near + far (or 4 * the row when the axis is not subsampled)
                const int sy = v2 ? y / 2 : y;
                const int far = v2 ? std::clamp(sy + ((y & 1) ? 1 : -1), 0, chh - 1) : sy;
                const uint8_t *pn = &comp[i].plane[static_cast<size_t>(std::min(sy, chh - 1)) * sw], *pf = &comp[i].plane[static_cast<size_t>(far) * sw];
                for (int x = 0; x < cw; x++)
                    row[static_cast<size_t>(x)] = v2 ? 3 * pn[x] + pf[x] : 4 * pn[x];
                for (int x = 0; x < width; x++)
                {
                    int v16;
                    if (h2)
                    {
                        const int sx = std::min(x / 2, cw - 1), fx = std::clamp(sx + ((x & 1) ? 1 : -1), 0, cw - 1);
                        v16 = 3 * row[static_cast<size_t>(sx)] + row[static_cast<size_t>(fx)]; // / 16
                    }
                    else
                        v16 = 4 * row[static_cast<size_t>(std::min(x, cw - 1))];
                    dst[static_cast<size_t>(y) * width + x] = static_cast<uint8_t>((v16 + 8) >> 4);
                }
            }
            (void)sh;
        }
        else
            for (int y = 0; y < height; y++)
                for (int x = 0; x < width; x++)
                    dst[static_cast<size_t>(y) * width + x] = comp[i].plane[static_cast<size_t>(y * comp[i].v / vmax) * sw + x * comp[i].h / hmax];
    }
    const bool ycc = ncomp == 3 && !(adobe && adobeTransform == 0);
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++)
        {
            uint8_t *o = &img.Pixels[(static_cast<size_t>(y) * width + x) * 4];
            int v[3] = { 0, 128, 128 };
            for (int i = 0; i < ncomp; i++)
                v[i] = full[static_cast<size_t>(i)][static_cast<size_t>(y) * width + x];
            if (ncomp == 1)
                o[0] = o[1] = o[2] = static_cast<uint8_t>(v[0]);
            else if (ycc)
            {
                const float Y = static_cast<float>(v[0]), cb = static_cast<float>(v[1]) - 128.0f, cr = static_cast<float>(v[2]) - 128.0f;
                auto clamp8 = [](float c) { const int q = static_cast<int>(std::floor(c + 0.5f)); return static_cast<uint8_t>(q < 0 ? 0 : (q > 255 ? 255 : q)); };
                o[0] = clamp8(Y + 1.402f * cr);
                o[1] = clamp8(Y - 0.344136f * cb - 0.714136f * cr);
                o[2] = clamp8(Y + 1.772f * cb);
            }
            else
            {
                o[0] = static_cast<uint8_t>(v[0]); o[1] = static_cast<uint8_t>(v[1]); o[2] = static_cast<uint8_t>(v[2]);
            }
            o[3] = 255;
        }
    return img;
}

}
