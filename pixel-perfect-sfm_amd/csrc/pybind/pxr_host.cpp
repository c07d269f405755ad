// pxr_host.cpp -- compiled host binding (pybind11) for the part of the drop-in path that touches one Python object per
// observation: the dump of a pycolmap-style Reconstruction + FeatureSet into the flat scene arrays pxr_ba_build_problem
// takes.  The reference's bundle adjustment walks colmap::Reconstruction in C++ behind its pybind11 module
// (pixsfm/bundle_adjustment/bindings.cc:36-51 -> bundle_optimizer.h:139-165,247-317; the GIL is released around the
// solve, keypoint_adjustment/bindings.cc:17-30); here the walk over the Python objects is C++ too (attribute reads through
// the CPython API, no interpreter loop), the problem construction and everything after it are the C-ABI of
// include/pixsfm_hip.h.  Works on any objects with the pycolmap attribute names (images[i].points2D[j].point3D_id /
// has_point3D(), points3D[p].track.elements[k].image_id / point2D_idx, camera_id): the stand-ins of
// pixsfm_amd/api/reconstruction.py here, pycolmap's own classes in a pixsfm installation.
//
// Built by csrc/Makefile (target `host`) into pixsfm_amd/_pxr_host.*.so; pixsfm_amd.api.bundle_adjustment._SceneDump uses
// it when it is importable and is tested against its own pure-Python form (tests/test_api_cpu.py).
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>

#include <cstdint>
#include <cstring>
#include <limits>
#include <unordered_map>
#include <vector>

namespace py = pybind11;

namespace {

// point3D_id of a Point2D as an index-able id, -1 when it has none: the stand-ins store -1, pycolmap 2^64 - 1
// (kInvalidPoint3DId), which does not fit a signed 64-bit integer
int64_t point3d_id(PyObject* q, PyObject* name) {
  PyObject* v = PyObject_GetAttr(q, name);
  if (!v) throw py::error_already_set();
  int overflow = 0;
  const long long id = PyLong_AsLongLongAndOverflow(v, &overflow);
  Py_DECREF(v);
  if (overflow) return -1;
  if (id == -1 && PyErr_Occurred()) throw py::error_already_set();
  return id < 0 ? -1 : (int64_t)id;
}

long long int_attr(PyObject* o, PyObject* name) {
  PyObject* v = PyObject_GetAttr(o, name);
  if (!v) throw py::error_already_set();
  const long long r = PyLong_AsLongLong(v);
  Py_DECREF(v);
  if (r == -1 && PyErr_Occurred()) throw py::error_already_set();
  return r;
}

template <typename T>
py::array_t<T> as_array(const std::vector<T>& v) {
  py::array_t<T> a((py::ssize_t)v.size());
  if (!v.empty()) std::memcpy(a.mutable_data(), v.data(), sizeof(T) * v.size());
  return a;
}

// images, points: the objects in ascending id; img_of: image_id -> position in `images`.
// Returns (p2d_ptr int64[n_img + 1], p2d_point3D_id int64[total] (-1: none), track_ptr int64[n_pts + 1],
//          track_image int32[elements] (position in `images`), track_p2d int32[elements]).
py::tuple scene_arrays(py::list images, py::list points, py::dict img_of) {
  py::str s_points2D("points2D"), s_pid("point3D_id"), s_track("track"), s_elements("elements"), s_image_id("image_id"),
      s_p2d("point2D_idx");
  std::vector<int64_t> p2d_ptr{0}, ids, track_ptr{0};
  std::vector<int32_t> track_image, track_p2d;
  for (py::handle im : images) {
    py::object p2 = im.attr(s_points2D);
    py::sequence seq = py::reinterpret_borrow<py::sequence>(p2);
    const py::ssize_t n = py::len(seq);
    ids.reserve(ids.size() + (size_t)n);
    PyObject* fast = PySequence_Fast(seq.ptr(), "points2D must be a sequence");
    if (!fast) throw py::error_already_set();
    PyObject** items = PySequence_Fast_ITEMS(fast);
    try {
      for (py::ssize_t k = 0; k < n; ++k) ids.push_back(point3d_id(items[k], s_pid.ptr()));
    } catch (...) { Py_DECREF(fast); throw; }
    Py_DECREF(fast);
    p2d_ptr.push_back((int64_t)ids.size());
  }
  for (py::handle pt : points) {
    py::object el = pt.attr(s_track).attr(s_elements);
    PyObject* fast = PySequence_Fast(el.ptr(), "track.elements must be a sequence");
    if (!fast) throw py::error_already_set();
    const py::ssize_t n = PySequence_Fast_GET_SIZE(fast);
    PyObject** items = PySequence_Fast_ITEMS(fast);
    try {
      for (py::ssize_t k = 0; k < n; ++k) {
        PyObject* key = PyObject_GetAttr(items[k], s_image_id.ptr());
        if (!key) throw py::error_already_set();
        PyObject* pos = PyDict_GetItemWithError(img_of.ptr(), key);      // borrowed
        Py_DECREF(key);
        if (!pos) { if (PyErr_Occurred()) throw py::error_already_set(); throw py::key_error("track element refers to an image the reconstruction lacks"); }
        track_image.push_back((int32_t)PyLong_AsLong(pos));
        track_p2d.push_back((int32_t)int_attr(items[k], s_p2d.ptr()));
      }
    } catch (...) { Py_DECREF(fast); throw; }
    Py_DECREF(fast);
    track_ptr.push_back((int64_t)track_image.size());
  }
  return py::make_tuple(as_array(p2d_ptr), as_array(ids), as_array(track_ptr), as_array(track_image), as_array(track_p2d));
}

// The patch object of every (image position, point2D index): patch_dicts[k] is the `patches` dict of image k's FeatureMap
// (sparse) or a single patch object (dense map: every keypoint resolves to it) or None.  Also returns the address of each
// patch's buffer (data_ptr(), 0 for objects without one: device-resident ArenaPatch) for pxr_arena_upload_gather.
py::tuple patches_of(py::list patch_dicts, py::list dense_flags, py::array_t<int32_t, py::array::c_style> obs_image,
                     py::array_t<int32_t, py::array::c_style> obs_p2d) {
  const py::ssize_t n = obs_image.size();
  if (obs_p2d.size() != n) throw py::value_error("obs_image / obs_p2d differ in length");
  py::list out(n);
  py::array_t<uint64_t> ptrs(n);
  uint64_t* pp = ptrs.mutable_data();
  const int32_t* oi = obs_image.data();
  const int32_t* oj = obs_p2d.data();
  py::str s_ptr("_ptr");
  const py::ssize_t n_img = py::len(patch_dicts);
  for (py::ssize_t o = 0; o < n; ++o) {
    if (oi[o] < 0 || oi[o] >= n_img) throw py::index_error("observation refers to an image out of range");
    PyObject* d = PyList_GET_ITEM(patch_dicts.ptr(), oi[o]);
    if (d == Py_None) throw py::key_error("no feature map for an image of the problem");
    PyObject* patch;
    if (PyObject_IsTrue(PyList_GET_ITEM(dense_flags.ptr(), oi[o]))) {
      patch = d;
    } else {
      PyObject* key = PyLong_FromLong(oj[o]);
      if (PyDict_Check(d)) {
        patch = PyDict_GetItemWithError(d, key);   // borrowed
        Py_XINCREF(patch);
      } else {                                     // any other mapping (UserDict, a lazy container): __getitem__
        patch = PyObject_GetItem(d, key);          // new reference; KeyError / TypeError of the container travel up
      }
      Py_DECREF(key);
      if (!patch) { if (PyErr_Occurred()) throw py::error_already_set(); throw py::key_error("no feature patch for an observation"); }
      PyList_SET_ITEM(out.ptr(), o, patch);        // steals the reference
      PyObject* a = PyObject_GetAttr(patch, s_ptr.ptr());
      if (a) { pp[o] = (uint64_t)PyLong_AsUnsignedLongLong(a); Py_DECREF(a); }
      else { PyErr_Clear(); pp[o] = 0; }
      continue;
    }
    Py_INCREF(patch);
    PyList_SET_ITEM(out.ptr(), o, patch);          // steals the reference
    PyObject* a = PyObject_GetAttr(patch, s_ptr.ptr());
    if (a) { pp[o] = (uint64_t)PyLong_AsUnsignedLongLong(a); Py_DECREF(a); }
    else { PyErr_Clear(); pp[o] = 0; }
  }
  return py::make_tuple(out, ptrs);
}

// One arena slot per DISTINCT host patch object of a problem (dense mode: all keypoints of an image share one patch):
// index[k] = slot of patch_list[k]; per slot the patch object, the address of its buffer, its corner and scale -- read from the
// patch's `_meta` tuple (ptr, corner x, corner y, scale x, scale y, (H, W, C, dtype)) that features.FeaturePatch keeps beside its
// numpy views; patches of different shape or dtype are refused (ValueError).
py::tuple gather_patches(py::list patch_list) {
  const py::ssize_t n = py::len(patch_list);
  py::array_t<int64_t> index(n);
  int64_t* idx = index.mutable_data();
  // open-addressing table object -> slot (a node-based std::unordered_map spent more time in malloc than in the walk)
  size_t cap = 16;
  while (cap < (size_t)n * 2) cap <<= 1;
  std::vector<PyObject*> keys(cap, nullptr);
  std::vector<int64_t> vals(cap, 0);
  auto hash = [cap](PyObject* p) { return (size_t)(((uintptr_t)p >> 4) * 0x9E3779B97F4A7C15ull) & (cap - 1); };
  std::vector<PyObject*> uniq_ptrs;
  std::vector<uint64_t> ptrs;
  std::vector<int32_t> corners;
  std::vector<double> scales;
  uniq_ptrs.reserve((size_t)n); ptrs.reserve((size_t)n); corners.reserve((size_t)n * 2); scales.reserve((size_t)n * 2);
  py::str s_meta("_meta");
  py::object layout;                      // (H, W, C, dtype) of the first patch: all must agree
  for (py::ssize_t k = 0; k < n; ++k) {
    PyObject* p = PyList_GET_ITEM(patch_list.ptr(), k);
    size_t h = hash(p);
    while (keys[h] && keys[h] != p) h = (h + 1) & (cap - 1);
    if (keys[h]) { idx[k] = vals[h]; continue; }
    PyObject* meta = PyObject_GetAttr(p, s_meta.ptr());
    if (!meta) throw py::error_already_set();
    if (!PyTuple_Check(meta) || PyTuple_GET_SIZE(meta) != 6) { Py_DECREF(meta); throw py::type_error("patch._meta must be (ptr, cx, cy, sx, sy, layout)"); }
    PyObject* lay = PyTuple_GET_ITEM(meta, 5);
    if (ptrs.empty()) { layout = py::reinterpret_borrow<py::object>(lay); }
    else if (lay != layout.ptr()) {
      // the entries of the layout tuples are usually the SAME objects (small integers, one dtype instance): identity first
      bool same_items = PyTuple_Check(lay) && PyTuple_Check(layout.ptr()) && PyTuple_GET_SIZE(lay) == PyTuple_GET_SIZE(layout.ptr());
      if (same_items)
        for (Py_ssize_t q = 0; q < PyTuple_GET_SIZE(lay); ++q)
          if (PyTuple_GET_ITEM(lay, q) != PyTuple_GET_ITEM(layout.ptr(), q)) { same_items = false; break; }
      if (!same_items) {
        const int same = PyObject_RichCompareBool(lay, layout.ptr(), Py_EQ);
        if (same != 1) { Py_DECREF(meta); if (same < 0) throw py::error_already_set(); throw py::value_error("patches of different shape or dtype"); }
      }
    }
    ptrs.push_back((uint64_t)PyLong_AsUnsignedLongLong(PyTuple_GET_ITEM(meta, 0)));
    corners.push_back((int32_t)PyLong_AsLong(PyTuple_GET_ITEM(meta, 1)));
    corners.push_back((int32_t)PyLong_AsLong(PyTuple_GET_ITEM(meta, 2)));
    scales.push_back(PyFloat_AsDouble(PyTuple_GET_ITEM(meta, 3)));
    scales.push_back(PyFloat_AsDouble(PyTuple_GET_ITEM(meta, 4)));
    Py_DECREF(meta);
    if (PyErr_Occurred()) throw py::error_already_set();
    const int64_t s = (int64_t)ptrs.size() - 1;
    keys[h] = p; vals[h] = s;
    idx[k] = s;
    uniq_ptrs.push_back(p);
  }
  py::list uniq((py::ssize_t)uniq_ptrs.size());
  for (size_t q = 0; q < uniq_ptrs.size(); ++q) { Py_INCREF(uniq_ptrs[q]); PyList_SET_ITEM(uniq.ptr(), (py::ssize_t)q, uniq_ptrs[q]); }
  py::array_t<int32_t> c({(py::ssize_t)ptrs.size(), (py::ssize_t)2});
  py::array_t<double> sc({(py::ssize_t)ptrs.size(), (py::ssize_t)2});
  if (!ptrs.empty()) {
    std::memcpy(c.mutable_data(), corners.data(), sizeof(int32_t) * corners.size());
    std::memcpy(sc.mutable_data(), scales.data(), sizeof(double) * scales.size());
  }
  return py::make_tuple(index, uniq, as_array(ptrs), c, sc);
}

// Slot of every object of `items` in the list `uniq` (by identity), -1 where it is not in it: the second consumer of a shared
// arena (features.SharedArena) finds its patches among the uploaded ones without a Python dict of a million ids.
py::array_t<int64_t> slots_of(py::list uniq, py::list items) {
  const py::ssize_t nu = py::len(uniq), n = py::len(items);
  size_t cap = 16;
  while (cap < (size_t)nu * 2) cap <<= 1;
  std::vector<PyObject*> keys(cap, nullptr);
  std::vector<int64_t> vals(cap, 0);
  auto hash = [cap](PyObject* p) { return (size_t)(((uintptr_t)p >> 4) * 0x9E3779B97F4A7C15ull) & (cap - 1); };
  for (py::ssize_t k = 0; k < nu; ++k) {
    PyObject* p = PyList_GET_ITEM(uniq.ptr(), k);
    size_t h = hash(p);
    while (keys[h] && keys[h] != p) h = (h + 1) & (cap - 1);
    if (!keys[h]) { keys[h] = p; vals[h] = k; }
  }
  py::array_t<int64_t> out(n);
  int64_t* o = out.mutable_data();
  for (py::ssize_t k = 0; k < n; ++k) {
    PyObject* p = PyList_GET_ITEM(items.ptr(), k);
    size_t h = hash(p);
    while (keys[h] && keys[h] != p) h = (h + 1) & (cap - 1);
    o[k] = keys[h] ? vals[h] : -1;
  }
  return out;
}

// has_patch flags of the sparse feature maps: patch_dicts[k] is the {keypoint id: patch} dict of image k (or None: nothing set
// for that image), p2d_ptr the first point2D slot of every image.  flags[p2d_ptr[k] + id] = 1 for every key inside the image's
// keypoint range (one PyDict_Next walk per image instead of a numpy array of the keys + isin).
py::array_t<uint8_t> patch_flags(py::list patch_dicts, py::array_t<int64_t, py::array::c_style | py::array::forcecast> p2d_ptr) {
  const py::ssize_t n_img = py::len(patch_dicts);
  if (p2d_ptr.ndim() != 1 || p2d_ptr.shape(0) != n_img + 1) throw py::value_error("p2d_ptr must have one entry per image plus one");
  const int64_t* ptr = p2d_ptr.data();
  py::array_t<uint8_t> flags((py::ssize_t)ptr[n_img]);
  uint8_t* f = flags.mutable_data();
  std::memset(f, 0, (size_t)ptr[n_img]);
  for (py::ssize_t k = 0; k < n_img; ++k) {
    PyObject* d = PyList_GET_ITEM(patch_dicts.ptr(), k);
    if (d == Py_None) continue;
    if (!PyDict_Check(d)) throw py::type_error("patch_dicts entries must be dict or None");
    const int64_t count = ptr[k + 1] - ptr[k];
    PyObject *key, *value;
    Py_ssize_t pos = 0;
    while (PyDict_Next(d, &pos, &key, &value)) {
      const long long id = PyLong_AsLongLong(key);
      if (id == -1 && PyErr_Occurred()) throw py::error_already_set();
      if (id >= 0 && id < count) f[ptr[k] + id] = 1;
    }
  }
  return flags;
}

}  // namespace

PYBIND11_MODULE(_pxr_host, m) {
  m.doc() = "compiled host side of the pixsfm_amd drop-in path: scene dump of pycolmap-style objects (see pxr_host.cpp)";
  m.def("scene_arrays", &scene_arrays, py::arg("images"), py::arg("points"), py::arg("img_of"));
  m.def("gather_patches", &gather_patches, py::arg("patch_list"));
  m.def("patch_flags", &patch_flags, py::arg("patch_dicts"), py::arg("p2d_ptr"));
  m.def("slots_of", &slots_of, py::arg("uniq"), py::arg("items"));
  m.def("patches_of", &patches_of, py::arg("patch_dicts"), py::arg("dense_flags"), py::arg("obs_image"), py::arg("obs_p2d"));
}
