// stub of <colmap/base/projection.h>: unused by base/src/irls_optim.h
