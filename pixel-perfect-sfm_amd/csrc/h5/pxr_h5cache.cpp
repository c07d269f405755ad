// pxr_h5cache.cpp -- native reader of pixsfm's dense-feature cache (include/pixsfm_h5.h), plain C++ on the HDF5 C
// library.  Follows the reference's reader (pixsfm/features/src/featuremanager.cc:20-40, featureset.cc:24-135,
// featuremap.cc:60-267, featurepatch.cc:181-268; written there with HighFive) for the layout pixsfm/extract.py:98-127
// and features/store_features.py write.  Host-only: built by g++ into libpixsfm_h5.so, separate from the HIP library.
#include <hdf5.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "pixsfm_h5.h"
#include "pixsfm_hip.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

struct Hid {   // closes whatever kind of identifier it holds
  hid_t id;
  explicit Hid(hid_t i = -1) : id(i) {}
  ~Hid() { reset(); }
  Hid(const Hid&) = delete;
  Hid& operator=(const Hid&) = delete;
  void reset(hid_t n = -1) {
    if (id >= 0) {
      switch (H5Iget_type(id)) {
        case H5I_FILE: H5Fclose(id); break;
        case H5I_GROUP: H5Gclose(id); break;
        case H5I_DATASET: H5Dclose(id); break;
        case H5I_DATASPACE: H5Sclose(id); break;
        case H5I_DATATYPE: H5Tclose(id); break;
        case H5I_ATTR: H5Aclose(id); break;
        default: break;
      }
    }
    id = n;
  }
  operator hid_t() const { return id; }
  bool ok() const { return id >= 0; }
};

bool ends_with(const std::string& s, const char* e) {
  const size_t n = strlen(e);
  return s.size() >= n && s.compare(s.size() - n, n, e) == 0;
}
bool is_image_key(const std::string& s) {   // util/src/misc.h:30-33
  return ends_with(s, ".png") || ends_with(s, ".jpeg") || ends_with(s, ".jpg") || ends_with(s, ".JPEG") || ends_with(s, ".JPG");
}

std::vector<std::string> list_names(hid_t group) {
  std::vector<std::string> names;
  H5G_info_t info;
  if (H5Gget_info(group, &info) < 0) return names;
  for (hsize_t i = 0; i < info.nlinks; ++i) {
    const ssize_t len = H5Lget_name_by_idx(group, ".", H5_INDEX_NAME, H5_ITER_INC, i, nullptr, 0, H5P_DEFAULT);
    if (len < 0) continue;
    std::string s((size_t)len + 1, '\0');
    H5Lget_name_by_idx(group, ".", H5_INDEX_NAME, H5_ITER_INC, i, &s[0], s.size(), H5P_DEFAULT);
    s.resize((size_t)len);
    names.push_back(s);
  }
  return names;
}

bool is_group(hid_t loc, const std::string& name) {
  H5O_info_t oi;
  if (H5Oget_info_by_name(loc, name.c_str(), &oi, H5P_DEFAULT) < 0) return false;
  return oi.type == H5O_TYPE_GROUP;
}

// GetImageKeys (util/src/misc.h:30-50)
void image_keys(hid_t group, const std::string& path, std::vector<std::string>* out) {
  for (const std::string& key : list_names(group)) {
    const std::string full = path.empty() ? key : path + "/" + key;
    if (is_image_key(key)) {
      out->push_back(full);
    } else if (is_group(group, key)) {
      Hid sub(H5Gopen2(group, key.c_str(), H5P_DEFAULT));
      if (sub.ok()) image_keys(sub, full, out);
    }
  }
}

int read_int_attr(hid_t obj, const char* name, std::vector<int>* out) {
  Hid a(H5Aopen(obj, name, H5P_DEFAULT));
  if (!a.ok()) return fail(PXR_EINVAL, "missing attribute '%s'", name);
  Hid sp(H5Aget_space(a));
  const hssize_t n = H5Sget_simple_extent_npoints(sp);
  out->assign((size_t)(n > 0 ? n : 0), 0);
  if (n > 0 && H5Aread(a, H5T_NATIVE_INT, out->data()) < 0) return fail(PXR_EINVAL, "cannot read attribute '%s'", name);
  return PXR_OK;
}

int read_string_attr(hid_t obj, const char* name, std::string* out) {
  Hid a(H5Aopen(obj, name, H5P_DEFAULT));
  if (!a.ok()) return fail(PXR_EINVAL, "missing attribute '%s'", name);
  Hid t(H5Aget_type(a));
  if (H5Tget_class(t) != H5T_STRING) return fail(PXR_EINVAL, "attribute '%s' is not a string", name);
  if (H5Tis_variable_str(t) > 0) {   // h5py stores str attributes as variable-length UTF-8
    Hid mt(H5Tcopy(H5T_C_S1));
    H5Tset_size(mt, H5T_VARIABLE);
    H5Tset_cset(mt, H5Tget_cset(t));
    char* s = nullptr;
    if (H5Aread(a, mt, &s) < 0 || !s) return fail(PXR_EINVAL, "cannot read attribute '%s'", name);
    *out = s;
    H5free_memory(s);
  } else {
    const size_t n = H5Tget_size(t);
    std::vector<char> buf(n + 1, '\0');
    if (H5Aread(a, t, buf.data()) < 0) return fail(PXR_EINVAL, "cannot read attribute '%s'", name);
    *out = buf.data();
  }
  return PXR_OK;
}

// IEEE binary16 the way h5py / numpy.float16 declare it
hid_t make_half_type() {
  hid_t t = H5Tcopy(H5T_IEEE_F32LE);
  H5Tset_fields(t, 15, 10, 5, 0, 10);
  H5Tset_size(t, 2);
  H5Tset_ebias(t, 15);
  return t;
}

struct MapShape {
  int format = 0, is_sparse = 1, H = 0, W = 0, C = 0;
  int64_t n = 0;
  bool windows = false;               // dense map stored once, loaded as sparse patch_size windows
  std::vector<int> ids;               // stored keypoint ids (format 2) / dataset names (format 1)
};

}  // namespace

struct pxr_h5 {
  Hid file;
  std::string prefix;
  std::vector<int> channels;
  int dtype = PXR_F16;
  std::vector<std::vector<std::string>> images;   // per level
  Hid half_type;
  hid_t mem_type() const { return dtype == PXR_F16 ? (hid_t)half_type : (dtype == PXR_F32 ? H5T_NATIVE_FLOAT : H5T_NATIVE_DOUBLE); }
  size_t elem() const { return dtype == PXR_F16 ? 2 : (dtype == PXR_F32 ? 4 : 8); }
};

namespace {

int open_map(pxr_h5* f, int level, const char* image, Hid* group) {
  if (!f || !image) return fail(PXR_EINVAL, "NULL argument");
  if (level < 0 || level >= (int)f->channels.size()) return fail(PXR_EINVAL, "level %d outside [0, %zu)", level, f->channels.size());
  const std::string path = "/" + f->prefix + std::to_string(level) + "/" + image;
  group->reset(H5Gopen2(f->file, path.c_str(), H5P_DEFAULT));
  if (!group->ok()) return fail(PXR_EINVAL, "no feature map '%s' in the cache", path.c_str());
  return PXR_OK;
}

// FeatureMap::InitFromH5Group (featuremap.cc:60-75) + InitFromH5GroupChunked (:134-215) / LoadFromH5Grouped (:92-132)
int map_shape(hid_t g, MapShape* s) {
  std::vector<int> v;
  if (int rc = read_int_attr(g, "format", &v)) return rc;
  if (v.size() != 1 || (v[0] != 1 && v[0] != 2)) return fail(PXR_EINVAL, "Unknown featuremap format.");   // featuremap.cc:73
  s->format = v[0];
  if (int rc = read_int_attr(g, "is_sparse", &v)) return rc;
  s->is_sparse = v.empty() ? 1 : (v[0] != 0);
  if (s->format == 2) {
    Hid d(H5Dopen2(g, "patches", H5P_DEFAULT)), k(H5Dopen2(g, "keypoint_ids", H5P_DEFAULT));
    if (!d.ok() || !k.ok()) return fail(PXR_EINVAL, "chunked feature map without 'patches' / 'keypoint_ids'");
    Hid sp(H5Dget_space(d));
    hsize_t dims[4] = {0, 0, 0, 0};
    if (H5Sget_simple_extent_ndims(sp) != 4 || H5Sget_simple_extent_dims(sp, dims, nullptr) < 0)
      return fail(PXR_EINVAL, "'patches' must be [n][H][W][C]");
    Hid ksp(H5Dget_space(k));
    const hssize_t nk = H5Sget_simple_extent_npoints(ksp);
    s->ids.assign((size_t)(nk > 0 ? nk : 0), 0);
    if (nk > 0 && H5Dread(k, H5T_NATIVE_INT, H5S_ALL, H5S_ALL, H5P_DEFAULT, s->ids.data()) < 0)
      return fail(PXR_EINVAL, "cannot read 'keypoint_ids'");
    s->H = (int)dims[1]; s->W = (int)dims[2]; s->C = (int)dims[3];
    s->n = (int64_t)s->ids.size();
    if (!s->is_sparse && s->ids.size() > 1) {   // "storing patch as dense but loading as sparse", featuremap.cc:157-165
      if (int rc = read_int_attr(g, "patch_size", &v)) return rc;
      if (v.size() != 1 || v[0] < 1 || v[0] > s->H || v[0] > s->W) return fail(PXR_EINVAL, "bad 'patch_size'");
      s->H = s->W = v[0];
      s->is_sparse = 1;
      s->windows = true;
    } else if ((int64_t)dims[0] != s->n) {
      return fail(PXR_EINVAL, "'patches' holds %lld entries, 'keypoint_ids' %lld", (long long)dims[0], (long long)s->n);
    }
  } else {
    for (const std::string& key : list_names(g)) {
      char* end = nullptr;
      const long id = strtol(key.c_str(), &end, 10);
      if (end && *end == '\0') s->ids.push_back((int)id);
    }
    s->n = (int64_t)s->ids.size();
    if (s->n > 0) {
      Hid d(H5Dopen2(g, std::to_string(s->ids[0]).c_str(), H5P_DEFAULT));
      if (!d.ok()) return fail(PXR_EINVAL, "cannot open patch dataset");
      Hid sp(H5Dget_space(d));
      hsize_t dims[3] = {0, 0, 0};
      if (H5Sget_simple_extent_ndims(sp) != 3 || H5Sget_simple_extent_dims(sp, dims, nullptr) < 0)
        return fail(PXR_EINVAL, "a grouped patch must be [H][W][C]");
      s->H = (int)dims[0]; s->W = (int)dims[1]; s->C = (int)dims[2];
    }
  }
  return PXR_OK;
}

int read_meta(hid_t g, const MapShape& s, std::vector<int>* corners, std::vector<double>* scales) {
  corners->assign((size_t)s.n * 2, 0);
  scales->assign((size_t)s.n * 2, 1.0);
  if (s.n == 0) return PXR_OK;
  if (s.format == 2) {
    Hid c(H5Dopen2(g, "corners", H5P_DEFAULT)), sc(H5Dopen2(g, "scales", H5P_DEFAULT));
    if (!c.ok() || !sc.ok()) return fail(PXR_EINVAL, "chunked feature map without 'corners' / 'scales'");
    Hid csp(H5Dget_space(c)), ssp(H5Dget_space(sc));
    if (H5Sget_simple_extent_npoints(csp) != (hssize_t)s.n * 2 || H5Sget_simple_extent_npoints(ssp) != (hssize_t)s.n * 2)
      return fail(PXR_EINVAL, "'corners' / 'scales' do not match 'keypoint_ids'");   // THROW_CHECK_EQ, featuremap.cc:193,199
    if (H5Dread(c, H5T_NATIVE_INT, H5S_ALL, H5S_ALL, H5P_DEFAULT, corners->data()) < 0 ||
        H5Dread(sc, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, scales->data()) < 0)
      return fail(PXR_EINVAL, "cannot read 'corners' / 'scales'");
  } else {
    for (int64_t i = 0; i < s.n; ++i) {   // FeaturePatch::LoadFromH5Dataset, featurepatch.cc:208-210
      Hid d(H5Dopen2(g, std::to_string(s.ids[(size_t)i]).c_str(), H5P_DEFAULT));
      if (!d.ok()) return fail(PXR_EINVAL, "cannot open patch dataset %d", s.ids[(size_t)i]);
      Hid ca(H5Aopen(d, "corner", H5P_DEFAULT)), sa(H5Aopen(d, "scale", H5P_DEFAULT));
      if (!ca.ok() || !sa.ok() || H5Aread(ca, H5T_NATIVE_INT, &(*corners)[2 * i]) < 0 ||
          H5Aread(sa, H5T_NATIVE_DOUBLE, &(*scales)[2 * i]) < 0)
        return fail(PXR_EINVAL, "patch %d without 'corner' / 'scale'", s.ids[(size_t)i]);
    }
  }
  return PXR_OK;
}

}  // namespace

extern "C" {

const char* pxr_h5_last_error(void) { return g_err; }

int pxr_h5_open(const char* path, const char* level_prefix, pxr_h5** out) {
  if (!path || !out) return fail(PXR_EINVAL, "pxr_h5_open: NULL argument");
  H5Eset_auto2(H5E_DEFAULT, nullptr, nullptr);   // errors are reported through return codes
  pxr_h5* f = new pxr_h5();
  f->file.reset(H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT));
  if (!f->file.ok()) { delete f; return fail(PXR_EINVAL, "pxr_h5_open: cannot open '%s'", path); }
  f->prefix = level_prefix ? level_prefix : "";
  f->half_type.reset(make_half_type());
  int rc = read_int_attr(f->file, "channels_per_level", &f->channels);   // featuremanager.cc:25-27
  std::string dt;
  if (!rc) rc = read_string_attr(f->file, "dtype", &dt);                // extract.py:100, :218-222
  if (!rc) {
    if (dt == "half") f->dtype = PXR_F16;
    else if (dt == "float") f->dtype = PXR_F32;
    else if (dt == "double") f->dtype = PXR_F64;
    else rc = fail(PXR_EINVAL, "pxr_h5_open: unknown dtype '%s'", dt.c_str());
  }
  for (size_t l = 0; !rc && l < f->channels.size(); ++l) {
    const std::string key = "/" + f->prefix + std::to_string(l);
    Hid g(H5Gopen2(f->file, key.c_str(), H5P_DEFAULT));
    if (!g.ok()) { rc = fail(PXR_EINVAL, "pxr_h5_open: no level group '%s'", key.c_str()); break; }
    f->images.emplace_back();
    image_keys(g, "", &f->images.back());
  }
  if (rc) { delete f; return rc; }
  *out = f;
  return PXR_OK;
}

int pxr_h5_close(pxr_h5* f) { delete f; return PXR_OK; }
int pxr_h5_num_levels(pxr_h5* f) { return f ? (int)f->channels.size() : 0; }
int pxr_h5_level_channels(pxr_h5* f, int level) { return (f && level >= 0 && level < (int)f->channels.size()) ? f->channels[(size_t)level] : -1; }
int pxr_h5_dtype(pxr_h5* f) { return f ? f->dtype : -1; }
int pxr_h5_num_images(pxr_h5* f, int level) { return (f && level >= 0 && level < (int)f->images.size()) ? (int)f->images[(size_t)level].size() : 0; }
const char* pxr_h5_image_name(pxr_h5* f, int level, int i) {
  if (!f || level < 0 || level >= (int)f->images.size() || i < 0 || i >= (int)f->images[(size_t)level].size()) return nullptr;
  return f->images[(size_t)level][(size_t)i].c_str();
}

int pxr_h5_map_info(pxr_h5* f, int level, const char* image, int* format, int* is_sparse, int64_t* n_patches, int* H, int* W, int* C) {
  Hid g;
  if (int rc = open_map(f, level, image, &g)) return rc;
  MapShape s;
  if (int rc = map_shape(g, &s)) return rc;
  if (format) *format = s.format;
  if (is_sparse) *is_sparse = s.is_sparse;
  if (n_patches) *n_patches = s.n;
  if (H) *H = s.H;
  if (W) *W = s.W;
  if (C) *C = s.C;
  return PXR_OK;
}

int pxr_h5_map_meta(pxr_h5* f, int level, const char* image, int32_t* keypoint_ids, int32_t* corners, double* scales) {
  Hid g;
  if (int rc = open_map(f, level, image, &g)) return rc;
  MapShape s;
  if (int rc = map_shape(g, &s)) return rc;
  std::vector<int> c;
  std::vector<double> sc;
  if (int rc = read_meta(g, s, &c, &sc)) return rc;
  for (int64_t i = 0; i < s.n; ++i) {
    if (keypoint_ids) keypoint_ids[i] = s.ids[(size_t)i];
    if (corners) { corners[2 * i] = c[2 * i]; corners[2 * i + 1] = c[2 * i + 1]; }
    if (scales) { scales[2 * i] = sc[2 * i]; scales[2 * i + 1] = sc[2 * i + 1]; }
  }
  return PXR_OK;
}

int pxr_h5_read_patches(pxr_h5* f, int level, const char* image, int64_t count, const int64_t* which, void* h_dst) {
  Hid g;
  if (int rc = open_map(f, level, image, &g)) return rc;
  MapShape s;
  if (int rc = map_shape(g, &s)) return rc;
  if (count < 0 || (count > 0 && !h_dst)) return fail(PXR_EINVAL, "pxr_h5_read_patches: invalid argument");
  const size_t patch_bytes = (size_t)s.H * s.W * s.C * f->elem();
  std::vector<int> corners;
  std::vector<double> scales;
  if (s.windows)
    if (int rc = read_meta(g, s, &corners, &scales)) return rc;
  Hid d, fsp;
  hsize_t fdims[4] = {0, 0, 0, 0};
  if (s.format == 2) {
    d.reset(H5Dopen2(g, "patches", H5P_DEFAULT));
    fsp.reset(H5Dget_space(d));
    H5Sget_simple_extent_dims(fsp, fdims, nullptr);
  }
  const hsize_t mdims[4] = {1, (hsize_t)s.H, (hsize_t)s.W, (hsize_t)s.C};
  Hid msp(H5Screate_simple(4, mdims, nullptr));
  for (int64_t q = 0; q < count; ++q) {
    const int64_t i = which ? which[q] : q;
    if (i < 0 || i >= s.n) return fail(PXR_EINVAL, "pxr_h5_read_patches: patch %lld outside [0, %lld)", (long long)i, (long long)s.n);
    char* dst = (char*)h_dst + patch_bytes * (size_t)q;
    if (s.format == 2) {   // LoadFromH5GroupChunked, featuremap.cc:236-260
      hsize_t start[4] = {(hsize_t)i, 0, 0, 0};
      if (s.windows) {
        const int cx = corners[2 * i], cy = corners[2 * i + 1];
        if (cx < 0 || cy < 0 || (hsize_t)(cy + s.H) > fdims[1] || (hsize_t)(cx + s.W) > fdims[2])
          return fail(PXR_EINVAL, "pxr_h5_read_patches: window of keypoint %d leaves the dense map", s.ids[(size_t)i]);
        start[0] = 0; start[1] = (hsize_t)cy; start[2] = (hsize_t)cx;
      }
      if (H5Sselect_hyperslab(fsp, H5S_SELECT_SET, start, nullptr, mdims, nullptr) < 0 ||
          H5Dread(d, f->mem_type(), msp, fsp, H5P_DEFAULT, dst) < 0)
        return fail(PXR_EINVAL, "pxr_h5_read_patches: cannot read patch %lld of '%s'", (long long)i, image);
    } else {               // FeaturePatch::LoadFromH5Dataset, featurepatch.cc:212-216
      Hid pd(H5Dopen2(g, std::to_string(s.ids[(size_t)i]).c_str(), H5P_DEFAULT));
      if (!pd.ok() || H5Dread(pd, f->mem_type(), H5S_ALL, H5S_ALL, H5P_DEFAULT, dst) < 0)
        return fail(PXR_EINVAL, "pxr_h5_read_patches: cannot read patch dataset %d of '%s'", s.ids[(size_t)i], image);
    }
  }
  return PXR_OK;
}

}  // extern "C"
