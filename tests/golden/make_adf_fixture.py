"""Fixture for the burn-in gate's Augmented Dickey-Fuller test (geonomics_amd/sim/burnin.py,
reference sim/burnin.py:81,94-96 -> statsmodels.tsa.stattools.adfuller).

statsmodels is not importable in this image (numpy 2), but a copy of statsmodels 0.12.2 sits
under /opt/conda with its data sets and its test suite.  This script stores the two DATA series
that suite tests adfuller on (US macro data, statsmodels/datasets/macrodata/macrodata.csv:
realgdp and infl, 203 quarters 1959Q1-2009Q3) together with the KNOWN ANSWERS the suite
asserts - values "taken from Stata" (statsmodels/tsa/tests/test_stattools.py:66-146) and the
autolag facts of tsa/tests/test_adfuller_lag.py:13-43:

  realgdp,       regression='c', autolag=None, maxlag=4: adfstat 0.97505319, p 0.99399563
  infl,          regression='c', autolag=None, maxlag=1: adfstat -4.3346988, p 0.00038661
  log(realgdp),  regression='c', autolag='aic', maxlag=None: 16 candidate lag lengths (maxlag 15),
                 usedlag 2, and (adfstat, p) equal to maxlag=2, autolag=None to 12 decimals

    python tests/golden/make_adf_fixture.py   ->  tests/golden/adf_macrodata.npz
"""
import csv
import os

import numpy as np

SRC = '/opt/conda/lib/python3.9/site-packages/statsmodels/datasets/macrodata/macrodata.csv'
rows = list(csv.DictReader(open(SRC)))
realgdp = np.array([float(r['realgdp']) for r in rows])
infl = np.array([float(r['infl']) for r in rows])
assert realgdp.size == 203
np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'adf_macrodata.npz'),
         realgdp=realgdp, infl=infl,
         # (series, maxlag, adfstat, pvalue): statsmodels 0.12.2 tsa/tests/test_stattools.py
         constant_realgdp=np.array([4, 0.97505319, 0.99399563]),
         constant_infl=np.array([1, -4.3346988, 0.00038661]),
         # tsa/tests/test_adfuller_lag.py: candidates, usedlag on log(realgdp)
         autolag_log_realgdp=np.array([16, 2]),
         meta=np.array('statsmodels 0.12.2 test-suite known answers (Stata); data: '
                       'statsmodels/datasets/macrodata/macrodata.csv'))
print('wrote adf_macrodata.npz')
