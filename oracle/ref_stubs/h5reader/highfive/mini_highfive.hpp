// Stand-in for the HighFive headers (third-party/HighFive is an EMPTY submodule in the reference checkout), written
// against the image's HDF5 C library so that the reference's own cache reader (features/src/featuremanager.cc,
// featureset.cc, featuremap.cc, featurepatch.cc, util/src/misc.h) compiles IN PLACE and reads real files
// (oracle/ref_h5_shim.cc).  Only the calls those files make exist here.  Semantics follow HighFive's: the memory type is
// the native type of the C++ destination (libhdf5 converts from the file type), vectors are resized to the extent of the
// dataspace, scalars accept a dataspace of exactly one element, half_float::half maps to a 2-byte IEEE float type
// (HIGHFIVE_USE_HALF_FLOAT, third-party/CMakeLists.txt:8).  Test infrastructure only.
#pragma once
#include <hdf5.h>

#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "third-party/half.hpp"

namespace HighFive {

struct Exception : std::runtime_error { using std::runtime_error::runtime_error; };
struct FileException : Exception { using Exception::Exception; };
struct GroupException : Exception { using Exception::Exception; };
struct DataSetException : Exception { using Exception::Exception; };
struct AttributeException : Exception { using Exception::Exception; };
struct DataSpaceException : Exception { using Exception::Exception; };

enum class ObjectType { File, Group, UserDataType, DataSpace, Dataset, Attribute, Other };

namespace detail {
struct Handle {   // shared ownership of one HDF5 identifier
  hid_t id;
  explicit Handle(hid_t i) : id(i) {}
  ~Handle() {
    if (id < 0) return;
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
};
using H = std::shared_ptr<Handle>;
inline H own(hid_t id) { return std::make_shared<Handle>(id); }

template <typename T> struct native;   // memory datatype of a C++ scalar (a NEW identifier the caller closes)
template <> struct native<int> { static hid_t make() { return H5Tcopy(H5T_NATIVE_INT); } };
template <> struct native<unsigned> { static hid_t make() { return H5Tcopy(H5T_NATIVE_UINT); } };
template <> struct native<long> { static hid_t make() { return H5Tcopy(H5T_NATIVE_LONG); } };
template <> struct native<unsigned long> { static hid_t make() { return H5Tcopy(H5T_NATIVE_ULONG); } };
template <> struct native<float> { static hid_t make() { return H5Tcopy(H5T_NATIVE_FLOAT); } };
template <> struct native<double> { static hid_t make() { return H5Tcopy(H5T_NATIVE_DOUBLE); } };
template <> struct native<half_float::half> {
  static hid_t make() {   // 1 sign, 5 exponent (bias 15), 10 mantissa bits
    hid_t t = H5Tcopy(H5T_NATIVE_FLOAT);
    H5Tset_fields(t, 15, 10, 5, 0, 10);
    H5Tset_size(t, 2);
    H5Tset_ebias(t, 15);
    return t;
  }
};

inline std::vector<size_t> dims_of(hid_t space) {
  const int nd = H5Sget_simple_extent_ndims(space);
  if (nd < 0) throw DataSpaceException("cannot read the dataspace rank");
  std::vector<hsize_t> d((size_t)nd);
  if (nd > 0) H5Sget_simple_extent_dims(space, d.data(), nullptr);
  return std::vector<size_t>(d.begin(), d.end());
}
inline size_t count_of(const std::vector<size_t>& d) { size_t n = 1; for (size_t x : d) n *= x; return n; }
// HighFive squeezes extents of one when it matches a dataspace with the rank of the destination
inline void check_rank(const std::vector<size_t>& d, size_t wanted, const char* what) {
  size_t rank = 0;
  for (size_t x : d) rank += (x != 1);
  if (wanted == 0 ? count_of(d) != 1 : rank > wanted)
    throw DataSpaceException(std::string("Impossible to read ") + what + ": the dataspace does not have the rank of the destination");
}
}  // namespace detail

class DataSpace {
 public:
  explicit DataSpace(std::vector<size_t> d = {}) : dims_(std::move(d)) {}
  std::vector<size_t> getDimensions() const { return dims_; }
  size_t getNumberDimensions() const { return dims_.size(); }
  size_t getElementCount() const { return detail::count_of(dims_); }
 private:
  std::vector<size_t> dims_;
};

class Attribute {
 public:
  explicit Attribute(detail::H h) : h_(std::move(h)) {}
  DataSpace getSpace() const { detail::Handle s(H5Aget_space(h_->id)); return DataSpace(detail::dims_of(s.id)); }
  template <typename T> void read(T* dst) const {
    detail::Handle t(detail::native<T>::make());
    if (H5Aread(h_->id, t.id, dst) < 0) throw AttributeException("Error during HDF5 Read of an attribute");
  }
  template <typename T> void read(T& dst) const { detail::check_rank(getSpace().getDimensions(), 0, "attribute"); read(&dst); }
  template <typename T> void read(std::vector<T>& dst) const {
    const auto d = getSpace().getDimensions();
    detail::check_rank(d, 1, "attribute");
    dst.resize(detail::count_of(d));
    read(dst.data());
  }
 private:
  detail::H h_;
};

class Selection {
 public:
  Selection(detail::H ds, detail::H filespace, std::vector<size_t> count) : ds_(std::move(ds)), fs_(std::move(filespace)), count_(std::move(count)) {}
  DataSpace getMemSpace() const { return DataSpace(count_); }
  template <typename T> void read(T* dst) const {
    std::vector<hsize_t> c(count_.begin(), count_.end());
    detail::Handle ms(H5Screate_simple((int)c.size(), c.data(), nullptr));
    detail::Handle t(detail::native<T>::make());
    if (H5Dread(ds_->id, t.id, ms.id, fs_->id, H5P_DEFAULT, dst) < 0) throw DataSetException("Error during HDF5 Read of a selection");
  }
 private:
  detail::H ds_, fs_;
  std::vector<size_t> count_;
};

class DataSet {
 public:
  explicit DataSet(detail::H h) : h_(std::move(h)) {}
  DataSpace getSpace() const { detail::Handle s(H5Dget_space(h_->id)); return DataSpace(detail::dims_of(s.id)); }
  std::vector<size_t> getDimensions() const { return getSpace().getDimensions(); }
  size_t getElementCount() const { return getSpace().getElementCount(); }
  Attribute getAttribute(const std::string& name) const {
    const hid_t a = H5Aopen(h_->id, name.c_str(), H5P_DEFAULT);
    if (a < 0) throw AttributeException("Unable to open the attribute \"" + name + "\"");
    return Attribute(detail::own(a));
  }
  template <typename T> void read(T* dst) const {
    detail::Handle t(detail::native<T>::make());
    if (H5Dread(h_->id, t.id, H5S_ALL, H5S_ALL, H5P_DEFAULT, dst) < 0) throw DataSetException("Error during HDF5 Read of a dataset");
  }
  template <typename T> void read(std::vector<T>& dst) const {
    const auto d = getDimensions();
    detail::check_rank(d, 1, "dataset");
    dst.resize(detail::count_of(d));
    read(dst.data());
  }
  Selection select(const std::vector<size_t>& offset, const std::vector<size_t>& count) const {
    const hid_t s = H5Dget_space(h_->id);
    std::vector<hsize_t> o(offset.begin(), offset.end()), c(count.begin(), count.end());
    if (s < 0 || H5Sselect_hyperslab(s, H5S_SELECT_SET, o.data(), nullptr, c.data(), nullptr) < 0) {
      if (s >= 0) H5Sclose(s);
      throw DataSpaceException("Unable to select hyperslab");
    }
    detail::H fs = detail::own(s);
    if (H5Sselect_valid(s) <= 0) throw DataSpaceException("Unable to select hyperslab: outside the extent of the dataset");
    return Selection(h_, fs, count);
  }
 private:
  detail::H h_;
};

class Group {
 public:
  explicit Group(detail::H h) : h_(std::move(h)) {}
  Attribute getAttribute(const std::string& name) const {
    const hid_t a = H5Aopen(h_->id, name.c_str(), H5P_DEFAULT);
    if (a < 0) throw AttributeException("Unable to open the attribute \"" + name + "\"");
    return Attribute(detail::own(a));
  }
  Group getGroup(const std::string& name) const {
    const hid_t g = H5Gopen2(h_->id, name.c_str(), H5P_DEFAULT);
    if (g < 0) throw GroupException("Unable to open the group \"" + name + "\"");
    return Group(detail::own(g));
  }
  DataSet getDataSet(const std::string& name) const {
    const hid_t d = H5Dopen2(h_->id, name.c_str(), H5P_DEFAULT);
    if (d < 0) throw DataSetException("Unable to open the dataset \"" + name + "\"");
    return DataSet(detail::own(d));
  }
  bool exist(const std::string& name) const { return H5Lexists(h_->id, name.c_str(), H5P_DEFAULT) > 0; }
  std::vector<std::string> listObjectNames() const {   // H5_INDEX_NAME, increasing (HighFive's default)
    std::vector<std::string> names;
    H5Literate(h_->id, H5_INDEX_NAME, H5_ITER_INC, nullptr,
               [](hid_t, const char* name, const H5L_info_t*, void* p) -> herr_t {
                 static_cast<std::vector<std::string>*>(p)->emplace_back(name);
                 return 0;
               }, &names);
    return names;
  }
  ObjectType getObjectType(const std::string& name) const {
    H5O_info_t info;
    if (H5Oget_info_by_name(h_->id, name.c_str(), &info, H5P_DEFAULT) < 0) throw GroupException("Unable to obtain info for \"" + name + "\"");
    switch (info.type) {
      case H5O_TYPE_GROUP: return ObjectType::Group;
      case H5O_TYPE_DATASET: return ObjectType::Dataset;
      case H5O_TYPE_NAMED_DATATYPE: return ObjectType::UserDataType;
      default: return ObjectType::Other;
    }
  }
 protected:
  detail::H h_;
};

class File : public Group {
 public:
  enum : unsigned { ReadOnly = 0x00u, ReadWrite = 0x01u };
  File(const std::string& path, unsigned flags = ReadOnly) : Group(open(path, flags)) {}
 private:
  static detail::H open(const std::string& path, unsigned flags) {
    const hid_t f = H5Fopen(path.c_str(), (flags & ReadWrite) ? H5F_ACC_RDWR : H5F_ACC_RDONLY, H5P_DEFAULT);
    if (f < 0) throw FileException("Unable to open file " + path);
    return detail::own(f);
  }
};

}  // namespace HighFive
