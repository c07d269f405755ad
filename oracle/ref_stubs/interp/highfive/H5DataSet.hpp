#pragma once
namespace HighFive { class DataSet; class Selection; class File; class Group; }
