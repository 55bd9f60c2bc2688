/* oracle/shim/R.h -- TEST INFRASTRUCTURE: empty stand-in (see RcppArmadillo.h in this directory) */
#pragma once
