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
        self._ptr = C.c_void_p()
        handle = (C.c_ubyte * 64)()
        nbytes = self.ring * self.world * self.capacity * 16
        with torch.cuda.device(self.device):
            L.call('tcvom_mbox_alloc', nbytes, C.byref(self._ptr), handle)
            bases = [0] * self.world
            bases[self.rank] = self._ptr.value
            if self.world > 1:
                me = (socket.gethostname(), os.getpid(), bytes(handle))
                infos = [None] * self.world
                dist.all_gather_object(infos, me, group=group)
                hosts = {i[0] for i in infos}
                if len(hosts) != 1:
                    self.close()
                    raise RuntimeError('PeerMailbox: the ranks span %d hosts; hipIpc mailboxes need one node' % len(hosts))
                for r, (_, pid, h) in enumerate(infos):
                    if r == self.rank:
                        continue
                    if pid == os.getpid():
                        raise RuntimeError('PeerMailbox: two ranks in one process')
                    if not any(h):
                        self.close()
                        raise RuntimeError('PeerMailbox: rank %d could not export its mailbox (hipIpcGetMemHandle)' % r)
                    p = C.c_void_p()
                    L.call('tcvom_mbox_open', (C.c_ubyte * 64).from_buffer_copy(h), C.byref(p))
                    self._opened.append(p)
                    bases[r] = p.value
            self.table = torch.tensor(bases, dtype=torch.int64, device=self.device)
        # pinned host word the kernels write on a timeout: read by the host without a synchronisation
        self.status = torch.zeros(4, dtype=torch.int32).pin_memory()
        self._sync = L.BnSync(peers=self.table.data_ptr(), world=self.world, rank=self.rank, seq=0, ring=self.ring,
                              capacity=self.capacity, timeout_ticks=self.timeout_ticks, status=self.status.data_ptr())
        if self.world > 1:
            dist.barrier(group=group)           # every peer has mapped every mailbox before the first push

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


def mailbox_for(group=None, device=None):
    """The PeerMailbox of (process group, device), created collectively on first use; None when the mailbox transport is not
    available (TCVOM_SYNCBN=rccl, ranks on several hosts, more than 32 ranks)."""
    if os.environ.get('TCVOM_SYNCBN', 'mailbox') != 'mailbox':
        return None
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    key = (id(group) if group is not None else None, dev)
    if key not in _MAILBOXES:
        world = dist.get_world_size(group)
        hosts = [None] * world
        dist.all_gather_object(hosts, socket.gethostname(), group=group)
        ok = len(set(hosts)) == 1 and world <= 32
        _MAILBOXES[key] = PeerMailbox(group, torch.device('cuda', dev)) if ok else None
    return _MAILBOXES[key]
