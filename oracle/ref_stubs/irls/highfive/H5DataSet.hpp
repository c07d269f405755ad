// stub of <highfive/H5DataSet.hpp>: unused by base/src/irls_optim.h
