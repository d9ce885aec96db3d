// image.cpp -- see image.h.  File formats follow src/image.cpp:20-111 byte for byte:
//   PPM  "P6\n<w> <h>\n255\n", rows bottom-up, channel = (unsigned char)(min(1, v) * 255)  (truncation)
//   HDR  "#?RADIANCE" header, "-Y h +X w", every scanline written as an "RLE" record that only uses
//        literal runs of <= 127 bytes, one colour plane after another, rows bottom-up
#include "image.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <limits>
#include <string>

namespace
{
inline float lum601(const Pixel& c) { return 0.299f * c.x + 0.587f * c.y + 0.114f * c.z; }
inline float clamp01(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }

struct Rgbe { uint8_t v[4]; };
// shared-exponent encoding: mantissas scaled by 256 / 2^e of the largest channel (image.cpp:55-69)
inline Rgbe to_rgbe(const Pixel& c)
{
    float d = std::max(std::max(c.x, c.y), c.z);
    if (d <= 1e-32) return Rgbe{{0, 0, 0, 0}};
    int   e;
    float m = std::frexp(d, &e);
    d       = m * 256.0f / d;
    return Rgbe{{static_cast<uint8_t>(c.x * d), static_cast<uint8_t>(c.y * d), static_cast<uint8_t>(c.z * d),
                 static_cast<uint8_t>(e + 128)}};
}
}  // namespace

Image::Image() { resize(0, 0); }
Image::Image(int w, int h) { resize(w, h); }
Image::~Image() {}

void Image::resize(int w, int h)
{
    m_width  = w;
    m_height = h;
    m_buffer.resize((size_t)w * h, Pixel(0.0f));
}

void Image::scale(float s)
{
    for (Pixel& p : m_buffer) { p.x *= s; p.y *= s; p.z *= s; p.w *= s; }
}

void Image::flip_updown()
{
    for (int r = 0; r < m_height / 2; r++)
        std::swap_ranges(m_buffer.begin() + (size_t)r * m_width, m_buffer.begin() + (size_t)(r + 1) * m_width,
                         m_buffer.begin() + (size_t)(m_height - 1 - r) * m_width);
}

void Image::accumulate_pixel(int i, int j, const Pixel& c)
{
    if (i < 0 || i >= m_width || j < 0 || j >= m_height) return;
    Pixel& p = m_buffer[i + (size_t)j * m_width];
    p.x += c.x; p.y += c.y; p.z += c.z;
}

void Image::accumulate_buffer(const Image& f)
{
    for (size_t n = 0; n < m_buffer.size(); n++)
    {
        m_buffer[n].x += f.m_buffer[n].x;
        m_buffer[n].y += f.m_buffer[n].y;
        m_buffer[n].z += f.m_buffer[n].z;
    }
}

void Image::tonemap_gamma(float gamma)
{
    const float ig = 1.0f / gamma;
    for (Pixel& p : m_buffer)
    {
        p.x = powf(clamp01(p.x), ig);
        p.y = powf(clamp01(p.y), ig);
        p.z = powf(clamp01(p.z), ig);
    }
}

// photoreceptor operator as configured by image.cpp:113-209: contrast m = 0.77, chromatic adaptation 0.5,
// light adaptation 0, intensity 0
void Image::tonemap_reinhard()
{
    const size_t n = m_buffer.size();
    if (!n) return;
    std::vector<float> lum(n);
    float l_mean = 0, r_mean = 0, g_mean = 0, b_mean = 0;
    for (size_t i = 0; i < n; i++)
    {
        lum[i] = std::max(lum601(m_buffer[i]), 1e-7f);
        l_mean += lum[i];
        r_mean += m_buffer[i].x;
        g_mean += m_buffer[i].y;
        b_mean += m_buffer[i].z;
    }
    l_mean /= n; r_mean /= n; g_mean /= n; b_mean /= n;
    const float m = 0.77f, c = 0.5f, a = 0.0f, f = expf(-0.0f);
    for (size_t i = 0; i < n; i++)
    {
        float ch[3]   = {m_buffer[i].x, m_buffer[i].y, m_buffer[i].z};
        float mean[3] = {r_mean, g_mean, b_mean};
        for (int k = 0; k < 3; k++)
        {
            float lc = c * ch[k] + (1.0f - c) * lum[i];
            float gc = c * mean[k] + (1.0f - c) * l_mean;
            float ca = a * lc + (1.0f - a) * gc;
            ch[k]    = ch[k] / (ch[k] + powf(f * ca, m));
        }
        m_buffer[i].x = ch[0]; m_buffer[i].y = ch[1]; m_buffer[i].z = ch[2];
    }
}

void Image::dump_ppm(const char* filename)
{
    FILE* fp = fopen(filename, "wb");
    if (!fp) { fprintf(stderr, "cannot write to %s\n", filename); return; }
    fprintf(fp, "P6\n%d %d\n255\n", m_width, m_height);
    std::vector<unsigned char> row((size_t)m_width * 3);
    for (int j = m_height - 1; j >= 0; --j)
    {
        for (int i = 0; i < m_width; ++i)
        {
            const Pixel& p = m_buffer[i + (size_t)j * m_width];
            row[3 * i + 0] = (unsigned char)(std::min(1.0f, p.x) * 255);
            row[3 * i + 1] = (unsigned char)(std::min(1.0f, p.y) * 255);
            row[3 * i + 2] = (unsigned char)(std::min(1.0f, p.z) * 255);
        }
        fwrite(row.data(), 1, row.size(), fp);
    }
    fclose(fp);
}

void Image::dump_hdr(const char* filename)
{
    FILE* fp = fopen(filename, "wb");
    if (!fp) { fprintf(stderr, "cannot write to %s\n", filename); return; }
    fprintf(fp, "#?RADIANCE\n# Made with custom writer\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=1.0\n\n-Y %d +X %d\n", m_height, m_width);
    std::vector<Rgbe>    line(m_width);
    std::vector<uint8_t> rec;
    for (int j = m_height - 1; j >= 0; --j)
    {
        for (int i = 0; i < m_width; i++) line[i] = to_rgbe(m_buffer[i + (size_t)j * m_width]);
        rec.clear();
        rec.push_back(2); rec.push_back(2);
        rec.push_back(uint8_t((m_width >> 8) & 0xFF)); rec.push_back(uint8_t(m_width & 0xFF));
        for (int k = 0; k < 4; k++)
            for (int cur = 0; cur < m_width;)
            {
                const int run = std::min(127, m_width - cur);
                rec.push_back(uint8_t(run));
                for (int i = cur; i < cur + run; i++) rec.push_back(line[i].v[k]);
                cur += run;
            }
        fwrite(rec.data(), 1, rec.size(), fp);
    }
    fclose(fp);
}

Pixel        Image::pixel(int i, int j) const { return m_buffer[i + (size_t)j * m_width]; }
int          Image::width() const { return m_width; }
int          Image::height() const { return m_height; }
const float* Image::buffer() const { return reinterpret_cast<const float*>(m_buffer.data()); }
float*       Image::buffer() { return reinterpret_cast<float*>(m_buffer.data()); }
