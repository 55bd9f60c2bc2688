/* oracle/shim/R_ext/Utils.h -- TEST INFRASTRUCTURE: empty stand-in (see RcppArmadillo.h in this directory) */
#pragma once
