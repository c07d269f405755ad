// stub of <highfive/H5Group.hpp>: unused by base/src/irls_optim.h
