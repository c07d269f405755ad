// ref_h5_shim.cc -- the reference's OWN reader of the dense-feature cache, compiled in place
// (/root/reference/pixsfm/features/src/featuremanager.cc, featureset.cc, featuremap.cc, featurepatch.cc and
// util/src/misc.h are on the compiler's command line / include path, oracle/Makefile target `ref`; nothing is copied), on
// the stand-in HighFive of ref_stubs/h5reader/highfive/mini_highfive.hpp (third-party/HighFive is an empty submodule) and
// the image's HDF5 C library.  The parity tests open the SAME file with this reader and with libpixsfm_h5.so and compare
// what each one hands out.  Test infrastructure only.
#include <algorithm>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "features/src/featuremanager.h"

namespace {
thread_local std::string g_err;

struct Base {
  virtual ~Base() {}
  virtual int num_levels() = 0;
  virtual int channels(int level) = 0;
  virtual std::vector<std::string> images(int level) = 0;
  virtual int map_info(int level, const std::string& image, int* is_sparse, int* n_patches, int* channels) = 0;
  virtual void patch_ids(int level, const std::string& image, unsigned* ids) = 0;
  virtual int patch(int level, const std::string& image, unsigned id, int* shape, int* corner, double* scale, int* has_data, int* refcount,
                    void* data, size_t capacity_bytes) = 0;
  virtual size_t load(int level, const std::string& image, const unsigned* ids, int n, int fill) = 0;
  virtual void unload(int level, const std::string& image, const unsigned* ids, int n) = 0;
  virtual size_t flush(int level) = 0;
};

template <typename dtype>
struct Reader : Base {
  std::unique_ptr<pixsfm::FeatureManager<dtype>> fm;
  Reader(std::string path, bool fill, std::string prefix) : fm(new pixsfm::FeatureManager<dtype>(path, fill, prefix)) {}
  int num_levels() override { return fm->NumLevels(); }
  int channels(int level) override { return fm->FeatureSet(level).Channels(); }
  std::vector<std::string> images(int level) override {
    std::vector<std::string> k = fm->FeatureSet(level).Keys();
    std::sort(k.begin(), k.end());
    return k;
  }
  int map_info(int level, const std::string& image, int* is_sparse, int* n_patches, int* channels) override {
    auto& m = fm->FeatureSet(level).GetFeatureMap(image);
    *is_sparse = m.IsSparse();
    *n_patches = m.NumPatches();
    *channels = m.Channels();
    return 0;
  }
  void patch_ids(int level, const std::string& image, unsigned* ids) override {
    std::vector<colmap::point2D_t> k = fm->FeatureSet(level).GetFeatureMap(image).Keys();
    std::sort(k.begin(), k.end());
    std::copy(k.begin(), k.end(), ids);
  }
  int patch(int level, const std::string& image, unsigned id, int* shape, int* corner, double* scale, int* has_data, int* refcount, void* data,
            size_t capacity_bytes) override {
    auto& m = fm->FeatureSet(level).GetFeatureMap(image);
    if (!m.HasFeaturePatch(id)) { g_err = "no such patch"; return -2; }
    auto& p = m.GetFeaturePatch(id);
    for (int i = 0; i < 3; ++i) shape[i] = p.Shape()[i];
    corner[0] = p.Corner()(0); corner[1] = p.Corner()(1);
    scale[0] = p.Scale()(0); scale[1] = p.Scale()(1);
    *has_data = p.HasData();
    *refcount = p.Status().reference_count;
    if (p.HasData() && data) {
      if (p.NumBytes() > capacity_bytes) { g_err = "destination too small"; return -3; }
      std::memcpy(data, p.Data(), p.NumBytes());
    }
    return 0;
  }
  size_t load(int level, const std::string& image, const unsigned* ids, int n, int fill) override {
    if (n < 0) {   // whole maps: FeatureSet::Load(required_maps, fill), featureset.cc:55-88
      std::unordered_set<std::string> req{image};
      return fm->FeatureSet(level).Load(req, fill != 0);
    }
    std::unordered_map<std::string, std::vector<colmap::point2D_t>> req;   // FeatureSet::Load(required_patches, fill), :90-143
    req[image].assign(ids, ids + n);
    return fm->FeatureSet(level).Load(req, fill != 0);
  }
  void unload(int level, const std::string& image, const unsigned* ids, int n) override {
    if (n < 0) {
      std::unordered_set<std::string> req{image};
      fm->FeatureSet(level).Unload(req);
      return;
    }
    std::unordered_map<std::string, std::vector<colmap::point2D_t>> req;
    req[image].assign(ids, ids + n);
    fm->FeatureSet(level).Unload(req);
  }
  size_t flush(int level) override { return fm->FeatureSet(level).Flush(); }
};

template <typename F>
auto guarded(F f, decltype(f()) on_error) -> decltype(f()) {
  try { return f(); }
  catch (const std::exception& e) { g_err = e.what(); }
  catch (...) { g_err = "unknown exception"; }
  return on_error;
}
}  // namespace

extern "C" {
const char* pxo_ref_h5_last_error() { return g_err.c_str(); }

// dtype: 0 half, 1 float, 2 double (the template argument the bindings pick from the file's "dtype" attribute,
// features/bindings.cc / extract.py load_features_from_cache)
void* pxo_ref_h5_open(const char* path, int fill, const char* level_prefix, int dtype) {
  H5Eset_auto2(H5E_DEFAULT, nullptr, nullptr);   // failures surface as exceptions of the stand-in, not as HDF5's stack dump
  return guarded([&]() -> void* {
    if (dtype == 0) return static_cast<Base*>(new Reader<half>(path, fill != 0, level_prefix));
    if (dtype == 1) return static_cast<Base*>(new Reader<float>(path, fill != 0, level_prefix));
    if (dtype == 2) return static_cast<Base*>(new Reader<double>(path, fill != 0, level_prefix));
    g_err = "dtype must be 0, 1 or 2";
    return nullptr;
  }, nullptr);
}
void pxo_ref_h5_close(void* h) { delete static_cast<Base*>(h); }
int pxo_ref_h5_num_levels(void* h) { return guarded([&] { return static_cast<Base*>(h)->num_levels(); }, -1); }
int pxo_ref_h5_channels(void* h, int level) { return guarded([&] { return static_cast<Base*>(h)->channels(level); }, -1); }
int pxo_ref_h5_num_images(void* h, int level) { return guarded([&] { return (int)static_cast<Base*>(h)->images(level).size(); }, -1); }
int pxo_ref_h5_image_name(void* h, int level, int i, char* out, int capacity) {
  return guarded([&] {
    const std::string s = static_cast<Base*>(h)->images(level).at(i);
    if ((int)s.size() + 1 > capacity) return -3;
    std::memcpy(out, s.c_str(), s.size() + 1);
    return 0;
  }, -1);
}
int pxo_ref_h5_map_info(void* h, int level, const char* image, int* is_sparse, int* n_patches, int* channels) {
  return guarded([&] { return static_cast<Base*>(h)->map_info(level, image, is_sparse, n_patches, channels); }, -1);
}
int pxo_ref_h5_patch_ids(void* h, int level, const char* image, unsigned* ids) {
  return guarded([&] { static_cast<Base*>(h)->patch_ids(level, image, ids); return 0; }, -1);
}
int pxo_ref_h5_patch(void* h, int level, const char* image, unsigned id, int* shape, int* corner, double* scale, int* has_data, int* refcount,
                     void* data, size_t capacity_bytes) {
  return guarded([&] { return static_cast<Base*>(h)->patch(level, image, id, shape, corner, scale, has_data, refcount, data, capacity_bytes); }, -1);
}
// n < 0: the whole map; returns the number of bytes the reference reports as loaded, or -1
long long pxo_ref_h5_load(void* h, int level, const char* image, const unsigned* ids, int n, int fill) {
  return guarded([&] { return (long long)static_cast<Base*>(h)->load(level, image, ids, n, fill); }, -1LL);
}
int pxo_ref_h5_unload(void* h, int level, const char* image, const unsigned* ids, int n) {
  return guarded([&] { static_cast<Base*>(h)->unload(level, image, ids, n); return 0; }, -1);
}
long long pxo_ref_h5_flush(void* h, int level) { return guarded([&] { return (long long)static_cast<Base*>(h)->flush(level); }, -1LL); }
}
