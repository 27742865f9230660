/*
 * tests/emu/emu.cpp -- TEST INFRASTRUCTURE: fiber scheduler behind tests/emu/emu.h.
 * x86-64 only (hand-written context switch); used solely by the CPU-side tests.
 */
#include "emu.h"
#include <stdlib.h>
#include <stdio.h>
#include <sys/mman.h>
#include <thread>
#include <vector>
#include <atomic>

extern "C" void emu_switch(void **from_sp, void **to_sp);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
	pushq %rbp
	pushq %rbx
	pushq %r12
	pushq %r13
	pushq %r14
	pushq %r15
	movq %rsp, (%rdi)
	movq (%rsi), %rsp
	popq %r15
	popq %r14
	popq %r13
	popq %r12
	popq %rbx
	popq %rbp
	ret
.size emu_switch,.-emu_switch
)");

namespace emu {

static const size_t STACK = 512 << 10;

struct Wave { uint64_t vals[2][64]; int arrived[2], left[2]; int nlive; uint64_t live_mask, arr_mask[2]; };
struct Block { int n; int nlive; int bar_arrived[2], bar_left[2]; Wave *waves; };
struct Fiber { void *sp; char *stack; bool done; unsigned tid; int phase, bphase; Block *blk; };

thread_local Fiber *cur;
thread_local uint3_t cur_tid, cur_bid, cur_bdim, cur_gdim;
thread_local char *dyn_lds;
static thread_local void *sched_sp;
static thread_local const std::function<void()> *cur_body;

static void trampoline()
{
	(*cur_body)();
	Fiber *f = cur;
	f->done = true;
	Wave &w = f->blk->waves[f->tid >> 6];
	--w.nlive; w.live_mask &= ~(1ull << (f->tid & 63));
	--f->blk->nlive;
	emu_switch(&f->sp, &sched_sp);
	abort();
}

void yield() { Fiber *f = cur; emu_switch(&f->sp, &sched_sp); }

void wave_exchange(uint64_t v, uint64_t out[64], uint64_t *live_mask)
{
	Fiber *f = cur; Wave &w = f->blk->waves[f->tid >> 6];
	int ph = f->phase, lane = f->tid & 63;
	w.vals[ph][lane] = v; ++w.arrived[ph]; w.arr_mask[ph] |= 1ull << lane;
	while (w.arrived[ph] < w.nlive) yield();
	memcpy(out, w.vals[ph], sizeof(w.vals[ph]));
	*live_mask = w.arr_mask[ph];   /* the participants of THIS rendezvous (lanes may exit right after it) */
	++w.left[ph];
	if (w.left[ph] == w.arrived[ph]) { w.arrived[ph] = w.left[ph] = 0; w.arr_mask[ph] = 0; }
	f->phase ^= 1;
}

void block_barrier()
{
	Fiber *f = cur; Block *b = f->blk; int ph = f->bphase;
	++b->bar_arrived[ph];
	while (b->bar_arrived[ph] < b->nlive) yield();
	++b->bar_left[ph];
	if (b->bar_left[ph] == b->bar_arrived[ph]) b->bar_arrived[ph] = b->bar_left[ph] = 0;
	f->bphase ^= 1;
}

static void run_block(unsigned bid, unsigned grid, unsigned block, std::vector<Fiber> &fib, std::vector<Wave> &waves)
{
	Block blk; memset(&blk, 0, sizeof(blk));
	blk.n = blk.nlive = (int)block; blk.waves = waves.data();
	unsigned nw = (block + 63) / 64;
	for (unsigned w = 0; w < nw; ++w) {
		memset(&waves[w], 0, sizeof(Wave));
		unsigned n = std::min(64u, block - w * 64);
		waves[w].nlive = (int)n; waves[w].live_mask = n == 64 ? ~0ull : ((1ull << n) - 1);
	}
	cur_bid = { bid, 0, 0 }; cur_bdim = { block, 1, 1 }; cur_gdim = { grid, 1, 1 };
	for (unsigned t = 0; t < block; ++t) {
		Fiber &f = fib[t];
		f.done = false; f.tid = t; f.phase = f.bphase = 0; f.blk = &blk;
		void **top = (void**)(f.stack + STACK);       /* 16-byte aligned */
		top[-1] = 0;                                    /* fake return slot: rsp%16==8 at entry */
		top[-2] = (void*)trampoline;
		for (int i = 3; i <= 8; ++i) top[-i] = 0;       /* rbp rbx r12 r13 r14 r15 */
		f.sp = &top[-8];
	}
	int remaining = (int)block;
	while (remaining) {
		int progressed = 0;
		for (unsigned t = 0; t < block; ++t) {
			Fiber &f = fib[t];
			if (f.done) continue;
			cur = &f; cur_tid = { t, 0, 0 };
			emu_switch(&sched_sp, &f.sp);
			if (f.done) { --remaining; ++progressed; }
		}
		(void)progressed;
	}
}

void launch(unsigned grid, unsigned block, size_t lds_bytes, const std::function<void()> &body)
{
	unsigned nthr = std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), grid);
	if (getenv("SSG_EMU_THREADS")) nthr = std::min<unsigned>((unsigned)atoi(getenv("SSG_EMU_THREADS")), grid);
	if (nthr < 1) nthr = 1;
	std::atomic<unsigned> next(0);
	auto worker = [&]() {
		std::vector<Fiber> fib(block); std::vector<Wave> waves((block + 63) / 64);
		char *stacks = (char*)mmap(0, STACK * block, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
		if (stacks == MAP_FAILED) { perror("mmap"); abort(); }
		for (unsigned t = 0; t < block; ++t) fib[t].stack = stacks + STACK * t;
		std::vector<char> lds(lds_bytes + 16);
		dyn_lds = lds.data();
		cur_body = &body;
		for (;;) {
			unsigned b = next.fetch_add(1);
			if (b >= grid) break;
			run_block(b, grid, block, fib, waves);
		}
		munmap(stacks, STACK * block);
	};
	if (nthr == 1) worker();
	else { std::vector<std::thread> th; for (unsigned i = 0; i < nthr; ++i) th.emplace_back(worker); for (auto &t : th) t.join(); }
}
}
