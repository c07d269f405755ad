// stub of <highfive/H5DataSpace.hpp>: unused by base/src/irls_optim.h
