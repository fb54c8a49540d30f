"""Burn-in stationarity tests (reference: geonomics/sim/burnin.py).

Host-side control logic, not a kernel: it decides WHEN genomes get assigned.
The reference calls statsmodels.tsa.stattools.adfuller (absent from this image,
un-pinned in the reference's requirements.txt) and scipy.stats.ttest_rel.
`adfuller` below restates statsmodels' published algorithm (regression='c',
autolag='AIC', maxlag = ceil(12 (n/100)^(1/4)), MacKinnon (1994) response-surface
p-values for N=1); scipy's ttest_rel is used directly.  Pinned since round 6 to the
known-answer values of statsmodels' own test suite (Stata's, on the US macro series
realgdp / infl): tests/golden/adf_macrodata.npz, tests/test_host_logic.py.
"""
import numpy as np
from scipy.stats import norm, ttest_rel

# MacKinnon (1994) approximate asymptotic p-value surfaces, regression 'c', N = 1
_TAU_MAX_C = 2.74
_TAU_MIN_C = -18.83
_TAU_STAR_C = -1.61
_TAU_C_SMALLP = np.array([2.1659, 1.4412, 3.8269e-2])
_TAU_C_LARGEP = np.array([1.7339, 9.3202e-1, -1.2745e-1, -1.0368e-2])


def mackinnonp(teststat):
    if teststat > _TAU_MAX_C:
        return 1.0
    if teststat < _TAU_MIN_C:
        return 0.0
    coef = _TAU_C_SMALLP if teststat <= _TAU_STAR_C else _TAU_C_LARGEP
    return float(norm.cdf(np.polyval(coef[::-1], teststat)))


def _lagmat_both_in(x, maxlag):
    """statsmodels lagmat(x[:, None], maxlag, trim='both', original='in'):
    column 0 = x[t], column k = x[t-k], rows t = maxlag .. n-1."""
    n = x.shape[0]
    cols = [x[maxlag - k:n - k] for k in range(0, maxlag + 1)]
    return np.column_stack(cols)


def _ols(y, X):
    pinv = np.linalg.pinv(X)
    beta = pinv @ y
    resid = y - X @ beta
    n, k = X.shape
    rank = np.linalg.matrix_rank(X)
    ssr = float(resid @ resid)
    df_resid = n - rank
    llf = -n / 2.0 * np.log(2 * np.pi) - n / 2.0 * np.log(ssr / n) - n / 2.0
    aic = -2 * llf + 2 * rank
    cov = pinv @ pinv.T * (ssr / df_resid) if df_resid > 0 else np.full((k, k), np.nan)
    bse = np.sqrt(np.diag(cov))
    with np.errstate(divide='ignore', invalid='ignore'):
        tvalues = beta / bse
    return beta, tvalues, aic


def adfuller(x, maxlag=None, autolag='AIC'):
    """Augmented Dickey-Fuller test, regression='c' (statsmodels.tsa.stattools.adfuller's
    defaults, which is how the reference calls it: sim/burnin.py:81,94-96); returns
    (adfstat, pvalue, usedlag, nobs).  maxlag=None: ceil(12 (n/100)^(1/4)); autolag='AIC'
    picks the lag length among 0..maxlag by AIC, autolag=None uses maxlag lags.  Pinned to
    statsmodels' own test-suite values (tests/test_host_logic.py, tests/golden/adf_macrodata.npz)."""
    x = np.asarray(x, dtype=float)
    if x.ndim != 1:
        raise ValueError('x must be 1d')
    if x.max() == x.min():
        raise ValueError('Invalid input, x is constant')
    if autolag not in (None, 'AIC', 'aic'):
        raise ValueError("autolag must be None or 'AIC'")
    nobs = x.shape[0]
    ntrend = 1
    if maxlag is None:
        maxlag = int(np.ceil(12.0 * np.power(nobs / 100.0, 1 / 4.0)))
        maxlag = min(nobs // 2 - ntrend - 1, maxlag)
        if maxlag < 0:
            raise ValueError('sample size is too short to use selected regression component')
    elif maxlag > nobs // 2 - ntrend - 1:
        raise ValueError('maxlag must be less than (nobs/2 - 1 - ntrend) where n trend is '
                         'the number of included deterministic regressors')
    xdiff = np.diff(x)
    xdall = _lagmat_both_in(xdiff, maxlag)
    nobs = xdall.shape[0]
    xdall[:, 0] = x[-nobs - 1:-1]
    xdshort = xdiff[-nobs:]
    if autolag is None:
        bestlag = maxlag
    else:
        # regressions on [const, level, dlag_1..dlag_m], m = 0..maxlag, over the SAME rows
        full = np.column_stack([np.ones(nobs), xdall])
        startlag = 2
        best_aic, bestlag = None, 0
        for lag in range(startlag, startlag + maxlag + 1):
            _, _, aic = _ols(xdshort, full[:, :lag])
            if best_aic is None or aic < best_aic:
                best_aic, bestlag = aic, lag - startlag
        # the chosen lag length is re-estimated on all the rows it leaves
        xdall = _lagmat_both_in(xdiff, bestlag)
        nobs = xdall.shape[0]
        xdall[:, 0] = x[-nobs - 1:-1]
        xdshort = xdiff[-nobs:]
    X = np.column_stack([xdall[:, :bestlag + 1], np.ones(nobs)])
    _, tvalues, _ = _ols(xdshort, X)
    adfstat = float(tvalues[0])
    return adfstat, mackinnonp(adfstat), bestlag, nobs


def _test_adf_threshold(spp, num_timesteps_back, alpha=0.05):
    """reference sim/burnin.py:94-96"""
    return adfuller(spp.Nt[-num_timesteps_back:])[1] < alpha


def _test_t_threshold(spp, num_timesteps_back, alpha=0.05):
    """reference sim/burnin.py:99-103"""
    num_timesteps_back += num_timesteps_back % 2
    return ttest_rel(spp.Nt[int(-num_timesteps_back): int(-num_timesteps_back / 2)],
                     spp.Nt[int(-num_timesteps_back / 2):])[1] > alpha


def spatial_test(stats, num_timesteps_back, alpha=0.05):
    """SpatialTester.run_test (reference sim/burnin.py:76-91) on the series of
    mean and std of per-cell count differences (computed on the device)."""
    results = []
    n = num_timesteps_back
    for key in ('mean', 'std'):
        data = stats[key]
        try:
            adf_res = adfuller(data[-n:])[1] < alpha
        except (ValueError, np.linalg.LinAlgError):
            adf_res = None
        try:
            ttest_res = ttest_rel(data[int(-n): int(-n / 2)], data[int(-n / 2):])[1] > alpha
        except ValueError:
            ttest_res = None
        results.append(adf_res and ttest_res)
    return bool(np.all(results))
