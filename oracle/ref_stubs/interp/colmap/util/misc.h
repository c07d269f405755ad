#pragma once
#include <cstring>
#include <string>
