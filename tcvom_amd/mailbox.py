"""SyncBatchNorm without collective calls: the peer mailbox behind `tcvom_bn_finalize_sync` / `tcvom_bn_bwd_finalize_sync`.

The reference converts every BatchNorm to `nn.SyncBatchNorm` before DDP (train_ddp.py:271-273): ~190 all_gathers + ~178
all_reduces of <= 4 KB per step, a chain of dependent, latency-bound collectives.  Here every rank owns
a mailbox of uncached device memory that its peers map through hipIpc; the BatchNorm finalize kernels push their fp64 sums
into the peers' mailboxes over xGMI and poll their own (csrc/norm.hip: bn_sync_exchange): no extra launch, no host
involvement, no collective library on the BatchNorm path.  RCCL / gloo all-reduce stays as the fallback (ranks on other
nodes, TCVOM_SYNCBN=rccl).

One `PeerMailbox` per process group; `tcvom_amd.ddp.convert_sync_batchnorm` hands it to the BatchNorm modules."""
import ctypes as C
import os
import socket

import torch
import torch.distributed as dist

from . import _lib as L

RING = 4                  # ring slots: an exchange may overwrite a slot `RING` exchanges later (2 suffice on one stream)
CAPACITY = 16384          # doubles per (slot, sender): frames x 2 x channels of one BatchNorm call (5 x 2 x 1280 = 12800)


class MailboxTimeout(RuntimeError):
    pass


class PeerMailbox(object):
    """world ranks of ONE node; rank r's mailbox is [RING][world][CAPACITY] pairs of 8-byte granules."""

    def __init__(self, group=None, device=None, ring=RING, capacity=CAPACITY, timeout_s=None, loopback=False):
        """loopback: a one-rank mailbox without a process group (the exchange runs against the rank's own memory: tests, and
        `bench.py --sync-bn` on one GPU, which measures the cost of the in-kernel exchange itself)."""
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.group = group
        if loopback:
            self.world, self.rank = 1, 0
        else:
            self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        if self.world > 32:
            raise ValueError('PeerMailbox: at most 32 ranks (one node), got %d' % self.world)
        self.ring, self.capacity = int(ring), int(capacity)
        t = float(os.environ.get('TCVOM_MBOX_TIMEOUT_S', '30')) if timeout_s is None else float(timeout_s)
        self.timeout_ticks = int(t * 1e8)
        self.seq = 0
        self.exchanges = 0                      # diagnostic: exchanges issued so far
        self._opened = []
        self.peer_devices = {}                  # rank -> HIP device ordinal the mapped peer mailbox reports (diagnostic)
        self._ptr = C.c_void_p()
        handle = (C.c_ubyte * 64)()
        nbytes = self.ring * self.world * self.capacity * 16
        with torch.cuda.device(self.device):
            try:
                L.call('tcvom_mbox_alloc', nbytes, C.byref(self._ptr), handle)
            except L.TcvomError:
                if self.world == 1:
                    raise
                self._ptr = C.c_void_p()        # publishes an all-zero handle below: every rank then raises together
            bases = [0] * self.world
            bases[self.rank] = self._ptr.value
            if self.world > 1:
                me = (socket.gethostname(), os.getpid(), bytes(handle))
                infos = [None] * self.world
                dist.all_gather_object(infos, me, group=group)
                hosts = {i[0] for i in infos}
                # (decided from the gathered data: every rank raises or none does)
                if len(hosts) != 1:
                    self.close()
                    raise RuntimeError('PeerMailbox: the ranks span %d hosts; hipIpc mailboxes need one node' % len(hosts))
                if len({i[1] for i in infos}) != self.world:
                    self.close()
                    raise RuntimeError('PeerMailbox: two ranks in one process')
                dead = [r for r, i in enumerate(infos) if not any(i[2])]
                if dead:
                    self.close()
                    raise RuntimeError('PeerMailbox: rank(s) %s could not export their mailbox (hipIpcGetMemHandle)' % dead)
                for r, (_, pid, h) in enumerate(infos):
                    if r == self.rank:
                        continue
                    p = C.c_void_p()
                    L.call('tcvom_mbox_open', (C.c_ubyte * 64).from_buffer_copy(h), C.byref(p))
                    self._opened.append(p)
                    bases[r] = p.value
                    d = C.c_int32(-1)
                    try:                                            # diagnostic only: which device the mapped memory reports
                        L.call('tcvom_mbox_device', p, C.byref(d))
                    except L.TcvomError:
                        pass
                    self.peer_devices[r] = int(d.value)
            self.table = torch.tensor(bases, dtype=torch.int64, device=self.device)
        # pinned host word the kernels write on a timeout: read by the host without a synchronisation
        self.status = torch.zeros(4, dtype=torch.int32).pin_memory()
        self.wait_ticks = torch.zeros(2, dtype=torch.int64, device=self.device)       # [max, sum] of the pull spin times (100 MHz ticks)
        self._sync = L.BnSync(peers=self.table.data_ptr(), world=self.world, rank=self.rank, seq=0, ring=self.ring,
                              capacity=self.capacity, timeout_ticks=self.timeout_ticks, status=self.status.data_ptr(),
                              wait_ticks=self.wait_ticks.data_ptr())
        # (every peer must have mapped every mailbox before the first push: mailbox_for() all-reduces a flag after construction;
        #  a direct user of this class calls dist.barrier() itself)

    def fits(self, nframes, channels):
        return nframes * 2 * channels <= self.capacity

    def next(self):
        """The tcvom_bn_sync of the next exchange (valid until the following call; the library copies it at launch)."""
        self.seq = (self.seq + 1) & 0xffffffff
        if self.seq == 0:
            self.seq = 1
        self.exchanges += 1
        if (self.exchanges & 255) == 0:
            self.check()
        self._sync.seq = self.seq
        return C.byref(self._sync)

    def self_test(self, channels=64, timeout_s=10.0):
        """One real exchange through the finalize kernel with known contributions (rank r contributes r + 1 to every channel's
        sum): every rank must read world (world + 1) / 2 -- i.e. it saw every peer's push with the right tag and data.  Uses
        (and consumes) one sequence number on every rank.  Returns False instead of hanging when a push never arrives."""
        dev = self.device
        saved_ticks = self._sync.timeout_ticks
        self._sync.timeout_ticks = int(timeout_s * 1e8)
        try:
            with torch.cuda.device(dev):
                K = channels
                part = torch.empty((1, 2, K), dtype=torch.float32, device=dev)
                part[0, 0] = float(self.rank + 1)
                part[0, 1] = float((self.rank + 1) ** 2)
                gamma = torch.ones(K, device=dev)
                beta = torch.zeros(K, device=dev)
                ss = torch.empty(2 * K, device=dev)
                saved = torch.empty(2 * K, device=dev)
                L.call('tcvom_bn_finalize_sync', L.ptr(part), 1, K, self.world, self.world, L.ptr(gamma), L.ptr(beta), 1e-5,
                       L.ptr(ss), L.ptr(saved), None, 1, 0, self.next(), L.stream_ptr())
                torch.cuda.synchronize(dev)
                want = (self.world + 1) / 2.0                                  # mean of 1 .. world
                ok = bool(torch.allclose(saved[:K], torch.full((K,), want, device=dev), rtol=0, atol=1e-6)) and int(self.status[0]) == 0
                self.status.zero_()
                self.exchanges -= 1                                            # (the diagnostic counts BatchNorm exchanges only)
                return ok
        finally:
            self._sync.timeout_ticks = saved_ticks

    def wait_stats(self, reset=True):
        """(longest pull in ms, sum of the pulls in ms) since the last reset: how long this rank's finalize kernels spun waiting
        for the next rank's push (one sample per exchange and workgroup).  Synchronises the device."""
        v = self.wait_ticks.tolist()
        if reset:
            self.wait_ticks.zero_()
        return v[0] / 1e5, v[1] / 1e5

    def crossed_devices(self):
        """True when a mapped peer mailbox lives on another HIP device than this rank's own (hipIpcOpenMemHandle crossed devices);
        None for a one-rank mailbox."""
        if not self.peer_devices:
            return None
        return any(d != self.device.index for d in self.peer_devices.values())

    def check(self):
        """Raise if a kernel gave up waiting for a peer (no synchronisation: reads the pinned status word)."""
        s = int(self.status[0])
        if s != 0:
            raise MailboxTimeout('SyncBatchNorm mailbox: exchange %d timed out waiting for a peer (rank %d of %d); the '
                                 'statistics of this step are invalid' % (s, self.rank, self.world))

    def close(self):
        for p in self._opened:
            try:
                L.call('tcvom_mbox_close', p)
            except L.TcvomError:
                pass
        self._opened = []
        if self._ptr and self._ptr.value:
            try:
                L.call('tcvom_mbox_free', self._ptr)
            except L.TcvomError:
                pass
            self._ptr = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_MAILBOXES = {}


def _all_ok(ok, group, device):
    """True when `ok` holds on EVERY rank (one small all-reduce)."""
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device if dist.get_backend(group) == 'nccl' else 'cpu')
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return bool(int(flag.item()))


def mailbox_for(group=None, device=None):
    """The PeerMailbox of (process group, device), created collectively on first use; None when the mailbox transport is not
    available or does not pass its self-test on every rank (then SyncBatchNorm uses one all-reduce per BatchNorm call):
    TCVOM_SYNCBN=rccl, ranks on several hosts, more than 32 ranks, hipIpc export / mapping refused, or peers' writes not
    observed.  Every decision is taken identically on all ranks (from gathered data or an all-reduced flag), so a failure on
    one rank never leaves the others waiting in a collective."""
    import warnings
    if os.environ.get('TCVOM_SYNCBN', 'mailbox') != 'mailbox':
        return None
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    key = (id(group) if group is not None else None, dev)
    if key in _MAILBOXES:
        return _MAILBOXES[key]
    cdev = torch.device('cuda', dev)
    world = dist.get_world_size(group)
    hosts = [None] * world
    dist.all_gather_object(hosts, socket.gethostname(), group=group)
    mb, why = None, None
    if len(set(hosts)) != 1 or world > 32:
        why = 'the ranks span %d hosts' % len(set(hosts)) if world <= 32 else 'more than 32 ranks'
    else:
        try:
            mb = PeerMailbox(group, cdev)          # (its constructor gathers the handles: every rank reaches that collective,
            ok = True                              #  a rank that cannot export publishes an all-zero handle and all ranks raise)
        except Exception as e:                     # noqa: BLE001 -- any local failure turns into a collective fallback
            mb, ok, why = None, False, 'setup failed on rank %d: %s' % (dist.get_rank(group), e)
        if not _all_ok(ok, group, cdev):
            if mb is not None:
                mb.close()
            mb, why = None, why or 'setup failed on another rank'
        else:
            try:
                ok = mb.self_test()
            except Exception as e:                 # noqa: BLE001
                ok, why = False, 'self-test raised on rank %d: %s' % (dist.get_rank(group), e)
            if not _all_ok(ok, group, cdev):
                mb.close()
                mb, why = None, why or 'self-test failed (a peer\'s pushes were not observed)'
    if mb is None and dist.get_rank(group) == 0:
        warnings.warn('SyncBatchNorm: peer mailboxes unavailable (%s); falling back to one all-reduce per BatchNorm call' % why)
    _MAILBOXES[key] = mb
    return mb
