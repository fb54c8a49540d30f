"""Wait for the worker processes of a multi-process test: the first non-zero exit ends
the others at once (a dead rank leaves its peers blocked in a collective), and no child
outlives the test whatever happens."""
import time


def wait_all(procs, timeout=600):
    t0 = time.time()
    rcs = [None] * len(procs)
    try:
        while any(rc is None for rc in rcs):
            for k, p in enumerate(procs):
                if rcs[k] is None:
                    rcs[k] = p.poll()
            if any(rc not in (None, 0) for rc in rcs):
                break
            if time.time() - t0 > timeout:
                raise TimeoutError('worker processes still running after %d s' % timeout)
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
                p.wait()
    return [p.returncode for p in procs]
