// stub of <highfive/H5File.hpp>: unused by base/src/irls_optim.h
