// image.h -- host-side float RGBA image that drains the accumulator: the capture path of the reference
// (host.cpp:585-610) hands the scaled / gamma-corrected frame to an Image and writes .ppm or .hdr.
// The member set is the one src/image.h:8-35 offers so that the reference's call sites compile unchanged;
// Pixel stands in for glm::vec4 (glm is not a dependency here).
#ifndef VOLPATH_HOST_IMAGE_H
#define VOLPATH_HOST_IMAGE_H
#include <vector>

struct Pixel
{
    float x = 0.0f, y = 0.0f, z = 0.0f, w = 0.0f;
    Pixel() = default;
    explicit Pixel(float v) : x(v), y(v), z(v), w(v) {}
    Pixel(float r, float g, float b, float a) : x(r), y(g), z(b), w(a) {}
};

class Image
{
    std::vector<Pixel> m_buffer;      // row-major, row 0 = bottom of the picture as rendered
    int                m_width = 0;
    int                m_height = 0;

public:
    Image();                          // empty image
    Image(int cols, int rows);        // zero-filled cols x rows
    ~Image();

    // ---- geometry and raw access
    int          width() const;
    int          height() const;
    Pixel        pixel(int col, int row) const;
    float*       buffer();            // 4 floats per pixel, what vp_download / cudaMemcpy fills
    const float* buffer() const;
    void         resize(int cols, int rows);   // new pixels are zero
    void         flip_updown();                // mirror the rows in place

    // ---- arithmetic on the stored radiance
    void scale(float factor);                                  // all four channels
    void accumulate_pixel(int col, int row, const Pixel& add); // rgb only; ignored outside the image
    void accumulate_buffer(const Image& other);                // rgb only; same size assumed

    // ---- tone mapping (in place, rgb only)
    void tonemap_gamma(float gamma);   // clamp to [0,1], then v^(1/gamma)
    void tonemap_reinhard();           // photoreceptor operator, contrast 0.77, chromatic adaptation 0.5

    // ---- file output; rows are written top-down, i.e. the stored row order reversed
    void dump_ppm(const char* path);   // binary P6, channel = trunc(min(1, v) * 255)
    void dump_hdr(const char* path);   // Radiance RGBE, scanline records of literal runs
};
#endif
