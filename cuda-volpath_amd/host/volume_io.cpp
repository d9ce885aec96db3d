#include "volume_io.h"

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

void* loadRawFile(const char* filename, size_t size)
{
    FILE* fp = fopen(filename, "rb");
    if (!fp) { fprintf(stderr, "Error opening file '%s'\n", filename); return nullptr; }
    void* data = malloc(std::max<size_t>(size, 1));
    if (!data) { fclose(fp); fprintf(stderr, "Out of memory reading '%s' (%zu bytes)\n", filename, size); return nullptr; }
    size_t got = fread(data, 1, size, fp);
    fclose(fp);
    printf("Read '%s', %zu bytes\n", filename, got);
    if (got != size)
    {
        // the reference returns the buffer with an uninitialised tail here (host.cpp:904-911); a short file is an error
        fprintf(stderr, "File '%s' is truncated: %zu of %zu bytes\n", filename, got, size);
        free(data);
        return nullptr;
    }
    return data;
}

void quantize_unit(const float* src, size_t n, VolumeType* dst)
{
    for (size_t i = 0; i < n; i++) dst[i] = VolumeType(std::max(0.0f, std::min(src[i], 1.0f)) * 255.0f);
}
void quantize_by_max(const float* src, size_t n, float max_value, VolumeType* dst)
{
    for (size_t i = 0; i < n; i++) dst[i] = VolumeType(std::max(0.0f, src[i]) / max_value * 255.0f);
}

void* loadBinaryFile(const char* filename, int& width, int& height, int& depth, bool quantized)
{
    FILE* fp = fopen(filename, "rb");
    if (!fp) { fprintf(stderr, "Error opening file '%s'\n", filename); return nullptr; }  // (the reference fclose()s NULL here)
    int32_t dims[3] = {-1, -1, -1};
    if (fread(dims, sizeof(int32_t), 3, fp) != 3) dims[0] = -1;
    width = dims[0]; height = dims[1]; depth = dims[2];
    if (width < 0 || height < 0 || depth < 0)
    {
        fclose(fp);
        fprintf(stderr, "Invalid resolution of file '%s'\n", filename);
        return nullptr;
    }
    size_t total = size_t(width) * size_t(height) * size_t(depth);
    if (total > (1llu << 33))
    {
        fclose(fp);
        fprintf(stderr, "Resolution too large of file '%s'\n", filename);
        return nullptr;
    }
    float* dataf = reinterpret_cast<float*>(malloc(sizeof(float) * std::max<size_t>(total, 1)));
    if (!dataf) { fclose(fp); fprintf(stderr, "Out of memory reading '%s'\n", filename); return nullptr; }
    size_t got = fread(dataf, sizeof(float), total, fp);
    fclose(fp);
    printf("Read '%s', %zu bytes\n", filename, got);
    if (got != total)
    {
        // a truncated dump would otherwise be quantised and uploaded with an uninitialised tail
        fprintf(stderr, "File '%s' is truncated: %zu of %zu voxels\n", filename, got, total);
        free(dataf);
        return nullptr;
    }
    if (!quantized) return dataf;
    VolumeType* data = reinterpret_cast<VolumeType*>(malloc(std::max<size_t>(total, 1)));
    if (!data) { free(dataf); fprintf(stderr, "Out of memory reading '%s'\n", filename); return nullptr; }
    quantize_unit(dataf, total, data);
    free(dataf);
    return data;
}

bool dump_dense_volume(const char* filename, const float* data, int nx, int ny, int nz)
{
    FILE* fp = fopen(filename, "wb");
    if (!fp) return false;
    int32_t dims[3] = {nx, ny, nz};
    bool ok = fwrite(dims, sizeof(int32_t), 3, fp) == 3;
    size_t n = (size_t)nx * ny * nz;
    ok = ok && fwrite(data, sizeof(float), n, fp) == n;
    fclose(fp);
    return ok;
}

#ifndef VOLPATH_WITH_OPENVDB
// host/vdb_openvdb.cpp holds the OpenVDB -> dense conversion; the Makefile builds it (and defines VOLPATH_WITH_OPENVDB) only
// where <openvdb/openvdb.h> exists.  The reference does not vendor OpenVDB either (vdbloader/CMakeLists.txt:2).
float* load_vdb(char* filename, int&, int&, int&, float&, float&)
{
    fprintf(stderr, "load_vdb('%s'): built without OpenVDB; convert the grid to the dense .bin dump and use loadBinaryFile\n", filename);
    return nullptr;
}
#endif

void* loadVdbFile(const char* filename, int& width, int& height, int& depth, bool quantized)
{
    FILE* probe = fopen(filename, "r");
    if (!probe) { fprintf(stderr, "Error opening file '%s'\n", filename); return nullptr; }
    fclose(probe);
    float min_value = 0, max_value = 0;
    float* dataf = load_vdb(const_cast<char*>(filename), width, height, depth, min_value, max_value);
    max_value    = std::max(max_value, 0.0001f);
    if (!dataf) { fprintf(stderr, "Error opening file '%s'\n", filename); return nullptr; }
    if (width < 0 || height < 0 || depth < 0)
    {
        fprintf(stderr, "Invalid resolution of file '%s'\n", filename);
        free(dataf);
        return nullptr;
    }
    size_t total = size_t(width) * size_t(height) * size_t(depth);
    if (total > (1llu << 33))
    {
        fprintf(stderr, "Resolution too large of file '%s'\n", filename);
        free(dataf);
        return nullptr;
    }
    if (!quantized) return dataf;
    VolumeType* data = reinterpret_cast<VolumeType*>(malloc(std::max<size_t>(total, 1)));
    quantize_by_max(dataf, total, max_value, data);
    free(dataf);
    return data;
}
