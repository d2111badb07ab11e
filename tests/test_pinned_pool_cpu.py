"""PinnedPool without a GPU: the allocator entry points replaced by libc malloc / free.  Checks the
hand-back of blocks by finalizers that run inside a garbage collection while the pool's lock is held
(advisor finding of round 3: a result kept alive only by a reference cycle used to dead-lock take())."""
import ctypes as C
import gc
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class _FakeLib(object):
    def __init__(self):
        self.libc = C.CDLL(None)
        self.libc.malloc.restype = C.c_void_p
        self.libc.malloc.argtypes = [C.c_size_t]
        self.libc.free.argtypes = [C.c_void_p]
        self.live = set()

    def cpol_host_alloc(self, ctx, size, out):
        p = self.libc.malloc(size)
        out._obj.value = p
        self.live.add(p)
        return 0

    def cpol_host_alloc_near(self, device, size, out):
        return self.cpol_host_alloc(None, size, out)

    def cpol_host_free(self, ctx, p):
        self.live.discard(p.value)
        self.libc.free(p)
        return 0


def _pool():
    from cosmo_pol_amd import _native as N
    pool = N.PinnedPool.__new__(N.PinnedPool)
    import collections
    pool.lib = _FakeLib()
    pool.device = None
    pool.free = {}
    pool.returned = collections.deque()
    pool.lock = threading.Lock()
    pool.closed = False
    pool.n_alloc = 0
    return pool


def test_block_of_a_result_in_a_reference_cycle_comes_back_through_a_collection_inside_take():
    pool = _pool()

    class Node(object):
        pass
    a, _ = pool.take(1000)
    n = Node()
    n.me, n.arr = n, a                      # the array is alive only through a cycle
    del a, n
    gc.disable()
    try:
        # a collection that runs while the lock is held, as an allocation inside take() may trigger it
        with pool.lock:
            gc.collect()                    # the finalizer runs HERE, on this thread: must not take the lock
        assert len(pool.returned) == 1
    finally:
        gc.enable()
    b, _ = pool.take(1000)                  # the same block again: nothing new allocated
    assert pool.n_alloc == 1
    del b
    pool.close()
    assert not pool.lib.live                # everything freed, also what finalizers hand back after close()
    c_alloc = pool.n_alloc
    assert c_alloc == 1


def test_blocks_released_after_close_are_freed_and_views_keep_the_block_alive():
    pool = _pool()
    a, _ = pool.take(3 << 20)
    view = a[100:200].view(np.float32)
    del a
    pool.close()
    assert len(pool.lib.live) == 1          # the view still owns the block
    view[:] = 1.0
    del view
    gc.collect()
    assert not pool.lib.live


def test_take_from_many_threads_with_collections_in_between():
    pool = _pool()
    errs = []

    def work():
        try:
            for i in range(200):
                a, _ = pool.take(4096)
                a[:8] = i % 250
                if i % 17 == 0:
                    gc.collect()
                del a
        except Exception as e:              # pragma: no cover
            errs.append(e)
    ts = [threading.Thread(target=work) for _ in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=60)
        assert not t.is_alive()
    assert not errs
    pool.close()
    gc.collect()
    assert not pool.lib.live
