/*
 * pixsfm_h5.h -- C-ABI of libpixsfm_h5.so: native reader of pixsfm's dense-feature cache (HDF5).
 *
 * SURVEY 8f row 2.  Replaces the HDF5 side of FeatureManager / FeatureSet / FeatureMap
 * (pixsfm/features/src/featuremanager.cc:20-40, featureset.cc:24-135, featuremap.cc:60-267; HighFive) for the file
 * layout written by pixsfm/extract.py:98-127 + features/store_features.py:
 *
 *   file attrs   "channels_per_level" int[n_levels], "dtype" "half" | "float" | "double"
 *   /<level_prefix><l>/<image name (may contain '/')>/          one group per image and level
 *       attrs    "format" (1 = grouped, 2 = chunked), "is_sparse" (int), + metadata ("patch_size", "scale", ...)
 *       format 2 "patches" [n][H][W][C], "keypoint_ids" int[n], "corners" int[n][2], "scales" double[n][2]
 *                (store_features.py:42-71; chunked one patch per chunk); a DENSE map stored with several keypoint
 *                ids is loaded as sparse patch_size windows at the stored corners (featuremap.cc:157-165,246-256)
 *       format 1 attr "shape", one dataset "<keypoint id>" [H][W][C] with attrs "corner" int[2], "scale" double[2]
 *                (store_features.py:17-39)
 *
 * Host-only code (plain C++ against the HDF5 C library of the image, /opt/conda/lib/libhdf5.so.103); the patches are
 * read straight into caller memory (e.g. a pinned staging buffer that pxr_arena_upload then sends to the device).
 * All functions return 0 or a PXR_E* code of pixsfm_hip.h; pxr_h5_last_error() describes the failure.
 */
#ifndef PIXSFM_H5_H_
#define PIXSFM_H5_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pxr_h5 pxr_h5;

const char* pxr_h5_last_error(void);

/* FeatureManager(h5_path, fill, level_prefix) (featuremanager.cc:20-40); the image names of every level are
 * enumerated like GetImageKeys (util/src/misc.h:30-50: nested groups, names ending in .png/.jpeg/.jpg/.JPEG/.JPG). */
int pxr_h5_open(const char* path, const char* level_prefix, pxr_h5** out);
int pxr_h5_close(pxr_h5* f);
int pxr_h5_num_levels(pxr_h5* f);
int pxr_h5_level_channels(pxr_h5* f, int level);
int pxr_h5_dtype(pxr_h5* f);                             /* PXR_F16 / PXR_F32 / PXR_F64 (file attr "dtype") */
int pxr_h5_num_images(pxr_h5* f, int level);
const char* pxr_h5_image_name(pxr_h5* f, int level, int i);

/* FeatureMap::InitFromH5Group (featuremap.cc:60-75,134-215): shape of one image's map.  *n_patches patches of
 * H x W x C as they will be LOADED (sparse windows for the dense-stored / sparse-loaded case), *is_sparse as the
 * reference's FeatureMap::IsSparse() reports it afterwards. */
int pxr_h5_map_info(pxr_h5* f, int level, const char* image, int* format, int* is_sparse, int64_t* n_patches,
                    int* H, int* W, int* C);
/* keypoint ids [n], corners [n][2] (x, y), scales [n][2] in stored order */
int pxr_h5_map_meta(pxr_h5* f, int level, const char* image, int32_t* keypoint_ids, int32_t* corners, double* scales);
/* LoadFromH5GroupChunked / LoadFromH5Grouped (featuremap.cc:92-132,217-267): patches which[0..count) (positions in
 * the stored order; NULL = 0 .. count-1) into h_dst [count][H][W][C] of the file's dtype. */
int pxr_h5_read_patches(pxr_h5* f, int level, const char* image, int64_t count, const int64_t* which, void* h_dst);

#ifdef __cplusplus
}
#endif
#endif /* PIXSFM_H5_H_ */
