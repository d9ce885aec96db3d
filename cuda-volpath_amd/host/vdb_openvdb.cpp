// vdb_openvdb.cpp -- load_vdb(): the first FloatGrid of an OpenVDB file as a dense float array (vdbloader/load_vdb.cpp:72-157).
//
// Compiled only on request (`make VOLPATH_WITH_OPENVDB=1`, -DVOLPATH_WITH_OPENVDB) where OpenVDB is installed.  The image this
// project is developed in has no OpenVDB: THIS FILE HAS NEVER BEEN COMPILED OR RUN.  It states the conversion the reference
// performs, against OpenVDB's public API:
//   * the grid that is converted is the first one in the file that is a FloatGrid (load_vdb.cpp:135-153);
//   * the dense array spans the bounding box of the ACTIVE voxels, evalActiveVoxelBoundingBox (:83-86), x fastest (:47-50);
//   * every active voxel -- active tiles included, voxel by voxel (the reference voxelises the topology mask, :99-101) -- gets
//     its value; everything else in the box stays 0 (the reference's std::vector is value-initialised, :31), NOT the grid's
//     background value;
//   * min / max are those of the active values, evalMinMax (:89-94).
#ifdef VOLPATH_WITH_OPENVDB
#include <openvdb/openvdb.h>

#include <cstdio>
#include <cstdlib>

#include "volume_io.h"

float* load_vdb(char* filename, int& width, int& height, int& depth, float& min_value, float& max_value)
{
    openvdb::initialize();
    openvdb::io::File file(filename);
    try { file.open(); }
    catch (const std::exception& e) { fprintf(stderr, "load_vdb('%s'): %s\n", filename, e.what()); return nullptr; }
    openvdb::GridPtrVecPtr grids = file.getGrids();
    file.close();
    openvdb::FloatGrid::ConstPtr grid;
    for (const auto& g : *grids)
        if ((grid = openvdb::gridPtrCast<openvdb::FloatGrid>(g))) break;
    if (!grid) { fprintf(stderr, "load_vdb('%s'): no float grid in the file\n", filename); return nullptr; }

    const openvdb::CoordBBox box = grid->evalActiveVoxelBoundingBox();
    const openvdb::Coord     lo = box.min(), dim = box.dim();
    if (box.empty() || dim.x() <= 0 || dim.y() <= 0 || dim.z() <= 0)
    {
        fprintf(stderr, "load_vdb('%s'): the float grid has no active voxel\n", filename);
        return nullptr;
    }
    width = dim.x(); height = dim.y(); depth = dim.z();
    grid->evalMinMax(min_value, max_value);
    const size_t nx = (size_t)dim.x(), ny = (size_t)dim.y(), total = nx * ny * (size_t)dim.z();
    float* dense = static_cast<float*>(calloc(total ? total : 1, sizeof(float)));
    if (!dense) { fprintf(stderr, "load_vdb('%s'): out of memory for %zu voxels\n", filename, total); return nullptr; }
    auto put = [&](const openvdb::Coord& c, float v) {
        const openvdb::Coord r = c - lo;
        dense[(size_t)r.x() + nx * ((size_t)r.y() + ny * (size_t)r.z())] = v;
    };
    for (auto it = grid->cbeginValueOn(); it; ++it)
    {
        if (it.isVoxelValue()) put(it.getCoord(), *it);
        else
        {
            openvdb::CoordBBox tile;
            it.getBoundingBox(tile);  // an active tile: every voxel of it carries the tile value
            tile.intersect(box);
            for (auto c = tile.begin(); c; ++c) put(*c, *it);
        }
    }
    return dense;
}
#endif
