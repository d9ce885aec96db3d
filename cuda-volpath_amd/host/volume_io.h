// volume_io.h -- dense-volume ingest with the reference's loader surface (host.cpp:895-1019,
// vdbloader/load_vdb.{h,cpp}): raw files, the dense ".bin" dump (int32 nx,ny,nz + float32[]),
// the [0,1] / max-normalised uchar quantisers, and load_vdb() behind its unchanged signature.
#ifndef VOLPATH_HOST_VOLUME_IO_H
#define VOLPATH_HOST_VOLUME_IO_H
#include <cstddef>

typedef unsigned char VolumeType;

// host.cpp:896-913: malloc'ed file contents (caller frees), nullptr on error
void* loadRawFile(const char* filename, size_t size);
// host.cpp:915-965: reads the dense dump; quantized -> uchar(clamp(v,0,1)*255) else the float array. malloc'ed.
void* loadBinaryFile(const char* filename, int& width, int& height, int& depth, bool quantized = true);
// host.cpp:968-1019: load_vdb + uchar(max(0,v)/max_value*255) quantiser
void* loadVdbFile(const char* filename, int& width, int& height, int& depth, bool quantized = true);
// vdbloader/load_vdb.h: first FloatGrid -> dense float array over the active bbox (+ dims and value range).
// host/vdb_openvdb.cpp implements it against OpenVDB; the Makefile compiles that file (defining VOLPATH_WITH_OPENVDB) only
// where <openvdb/openvdb.h> is found -- NOT in the image this project is developed in, where that file has therefore never
// been compiled.  Without it load_vdb reports the missing dependency and returns nullptr (the reference does not vendor
// OpenVDB either).
float* load_vdb(char* filename, int& width, int& height, int& depth, float& min_value, float& max_value);
// the dump format of vdbloader/load_vdb.cpp:52-69 (Volume::dump): int32 nx,ny,nz then nx*ny*nz float32, x fastest
bool dump_dense_volume(const char* filename, const float* data, int nx, int ny, int nz);
// the two quantisers on their own (host.cpp:955, :1009)
void quantize_unit(const float* src, size_t n, VolumeType* dst);
void quantize_by_max(const float* src, size_t n, float max_value, VolumeType* dst);
#endif
