"""MI355X-native LS-SPA (drop-in for cvxgrp/ls-spa's ``ls_spa`` package)."""
