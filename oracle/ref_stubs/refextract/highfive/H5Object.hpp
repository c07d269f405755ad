// Declarations for the HighFive names util/src/misc.h's GetImageKeys mentions (never called by the reference extractor).
#pragma once
#include <string>
#include <vector>
namespace HighFive {
class DataSet;
class Selection;
enum class ObjectType { File, Group, UserDataType, DataSpace, Dataset, Attribute, Other };
class Group {
 public:
  std::vector<std::string> listObjectNames() const;
  ObjectType getObjectType(const std::string&) const;
  Group getGroup(const std::string&) const;
};
}  // namespace HighFive
