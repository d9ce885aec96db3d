// image.h -- float RGBA image with the reference's Image surface (src/image.h:8-35): PPM / Radiance-HDR
// writers, gamma and Reinhard tone mapping, accumulate / scale / flip.  Pixel replaces glm::vec4.
#ifndef VOLPATH_HOST_IMAGE_H
#define VOLPATH_HOST_IMAGE_H
#include <vector>

struct Pixel
{
    float x = 0.0f, y = 0.0f, z = 0.0f, w = 0.0f;
    Pixel() = default;
    explicit Pixel(float v) : x(v), y(v), z(v), w(v) {}
    Pixel(float x_, float y_, float z_, float w_) : x(x_), y(y_), z(z_), w(w_) {}
};

class Image
{
public:
    Image();
    Image(int w, int h);
    ~Image();

    void resize(int w, int h);
    void scale(float s);
    void flip_updown();

    void accumulate_pixel(int i, int j, const Pixel& c);
    void accumulate_buffer(const Image& f);

    void tonemap_gamma(float gamma);
    void tonemap_reinhard();
    void dump_ppm(const char* filename);
    void dump_hdr(const char* filename);

    Pixel        pixel(int i, int j) const;
    int          width() const;
    int          height() const;
    const float* buffer() const;
    float*       buffer();

private:
    std::vector<Pixel> m_buffer;
    int                m_width, m_height;
};
#endif
