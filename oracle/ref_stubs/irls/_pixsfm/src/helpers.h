// stub: base/src/irls_optim.h includes it but uses nothing of it
