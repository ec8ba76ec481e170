// zra_amd — the optimal parsers of zstd 1.4.9 (btopt / btultra / btultra2, levels 13-22), bit-exact, one lane per frame.
// Restated from zstd_opt.c through oracle/zo_encode.c (mf_opt, opt_get_all_matches, opt_insert_bt1 and the price model), which is
// pinned byte for byte against libzstd; the statements below follow the oracle's one for one. Coverage, not speed: the forward parse
// is a serial dynamic program over up to 4096 positions with a binary-tree search at every one of them.
// Included by zra_encode_mf.hip inside its anonymous namespace (uses hashN, count_eq, hb32, ld32, Emit).
#pragma once

__constant__ u8 o_LLcode[64] = {0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,16,17,17,18,18,19,19,20,20,20,20,21,21,21,21,22,22,22,22,22,22,22,22,
                                23,23,23,23,23,23,23,23,24,24,24,24,24,24,24,24,24,24,24,24,24,24,24,24};
__constant__ u8 o_MLcode[128] = {0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,32,33,33,34,34,35,35,
                                 36,36,36,36,37,37,37,37,38,38,38,38,38,38,38,38,39,39,39,39,39,39,39,39,40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,
                                 41,41,41,41,41,41,41,41,41,41,41,41,41,41,41,41,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,
                                 42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42};
__constant__ u8 o_LLbits[36] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,6,7,8,9,10,11,12,13,14,15,16};
__constant__ u8 o_MLbits[53] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,4,5,7,8,9,10,11,12,13,14,15,16};

struct OptCtx {
  ZraOptState* o;
  u32* hashT; u32* bt; u32* hash3;
  u32 hashLog, chainLog, searchLog, minMatchParam, targetLength, hashLog3;
  u32 idxShift, nextToUpdate, windowLog;
  int lvl;

  __device__ static u32 ll_code(u32 v) { return v > 63 ? hb32(v) + 19 : o_LLcode[v]; }
  __device__ static u32 ml_code(u32 ml) { const u32 b = ml - 3; return b > 127 ? hb32(b) + 36 : o_MLcode[b]; }
  __device__ static u32 bit_weight(u32 stat) { return hb32(stat + 1) * 256u; }
  __device__ static u32 frac_weight(u32 raw) { const u32 stat = raw + 1, hb = hb32(stat); return hb * 256u + ((stat << 8) >> hb); }
  __device__ u32 weight(u32 stat) const { return lvl ? frac_weight(stat) : bit_weight(stat); }
  __device__ void set_base_prices() {
    o->litSumBasePrice = weight(o->litSum); o->litLengthSumBasePrice = weight(o->litLengthSum);
    o->matchLengthSumBasePrice = weight(o->matchLengthSum); o->offCodeSumBasePrice = weight(o->offCodeSum);
  }
  __device__ static u32 downscale(u32* t, u32 last, int malus) {
    u32 sum = 0;
    for (u32 s = 0; s <= last; s++) { t[s] = 1 + (t[s] >> (4 + malus)); sum += t[s]; }
    return sum;
  }
  __device__ static u32 upscale(u32* t, u32 last) {
    u32 sum = 0;
    for (u32 s = 0; s <= last; s++) { t[s] <<= 4; t[s]--; sum += t[s]; }
    return sum;
  }
  __device__ void upscale_stats() {
    o->litSum = upscale(o->litFreq, 255); o->litLengthSum = upscale(o->litLengthFreq, 35);
    o->matchLengthSum = upscale(o->matchLengthFreq, 52); o->offCodeSum = upscale(o->offCodeFreq, 31);
  }
  __device__ void rescale_freqs(const u8* blk, u32 n) {
    o->predef = 0;
    if (o->litLengthSum == 0) {
      if (n <= 1024) o->predef = 1;
      for (u32 k = 0; k < 256; k++) o->litFreq[k] = 0;
      for (u32 i = 0; i < n; i++) o->litFreq[blk[i]]++;
      o->litSum = downscale(o->litFreq, 255, 1);
      for (u32 k = 0; k <= 35; k++) o->litLengthFreq[k] = 1;
      o->litLengthSum = 36;
      for (u32 k = 0; k <= 52; k++) o->matchLengthFreq[k] = 1;
      o->matchLengthSum = 53;
      for (u32 k = 0; k <= 31; k++) o->offCodeFreq[k] = 1;
      o->offCodeSum = 32;
    } else {
      o->litSum = downscale(o->litFreq, 255, 1);
      o->litLengthSum = downscale(o->litLengthFreq, 35, 0);
      o->matchLengthSum = downscale(o->matchLengthFreq, 52, 0);
      o->offCodeSum = downscale(o->offCodeFreq, 31, 0);
    }
    set_base_prices();
  }
  __device__ u32 raw_literals_cost(const u8* lit, u32 n) const {
    if (n == 0) return 0;
    if (o->predef) return (n * 6) * 256u;
    u32 price = n * o->litSumBasePrice;
    for (u32 u = 0; u < n; u++) price -= weight(o->litFreq[lit[u]]);
    return price;
  }
  __device__ u32 ll_price(u32 ll) const {
    if (o->predef) return weight(ll);
    const u32 code = ll_code(ll);
    return (o_LLbits[code] * 256u) + o->litLengthSumBasePrice - weight(o->litLengthFreq[code]);
  }
  __device__ u32 match_price(u32 offset, u32 ml) const {
    const u32 offCode = hb32(offset + 1), mlBase = ml - 3;
    if (o->predef) return weight(mlBase) + ((16 + offCode) * 256u);
    u32 price = (offCode * 256u) + (o->offCodeSumBasePrice - weight(o->offCodeFreq[offCode]));
    if (lvl < 2 && offCode >= 20) price += (offCode - 19) * 2 * 256u;
    const u32 mlCode = ml_code(ml);
    price += (o_MLbits[mlCode] * 256u) + (o->matchLengthSumBasePrice - weight(o->matchLengthFreq[mlCode]));
    price += 256u / 5;
    return price;
  }
  __device__ void update_stats(u32 ll, const u8* lit, u32 offCode, u32 ml) {
    for (u32 u = 0; u < ll; u++) o->litFreq[lit[u]] += 2;
    o->litSum += ll * 2;
    o->litLengthFreq[ll_code(ll)]++; o->litLengthSum++;
    o->offCodeFreq[hb32(offCode + 1)]++; o->offCodeSum++;
    o->matchLengthFreq[ml_code(ml)]++; o->matchLengthSum++;
  }
  __device__ static void update_rep(u32* out, const u32* rep, u32 offset, u32 ll0) {
    if (offset >= 3) { const u32 r0 = rep[0], r1 = rep[1]; out[2] = r1; out[1] = r0; out[0] = offset - 2; }
    else {
      const u32 repCode = offset + ll0;
      if (repCode > 0) {
        const u32 cur = repCode == 3 ? rep[0] - 1 : rep[repCode];
        const u32 r2 = repCode >= 2 ? rep[1] : rep[2], r1 = rep[0];
        out[2] = r2; out[1] = r1; out[0] = cur;
      } else { const u32 a = rep[0], b = rep[1], cc = rep[2]; out[0] = a; out[1] = b; out[2] = cc; }
    }
  }
  __device__ static u32 hash3of(const u8* p, u32 h) { return ((ld32(p) << 8) * 506832829u) >> (32 - h); }
  __device__ u32 tmpl_mls() const { return minMatchParam < 4 ? (minMatchParam == 3 ? 3u : 4u) : minMatchParam > 6 ? 6u : minMatchParam; }

  // ZSTD_insertBt1
  __device__ u32 insert_bt1(const u8* src, u32 curr, u32 iend, u32 mls) {
    const u32 btMask = (1u << (chainLog - 1)) - 1;
    const u32 ipos = curr - 1 - idxShift;
    const u32 h = hashN(src + ipos, hashLog, mls);
    u32 matchIndex = hashT[h];
    u32 commonSmaller = 0, commonLarger = 0;
    const u32 btLow = btMask >= curr ? 0 : curr - btMask;
    u32* smallerPtr = bt + 2 * (curr & btMask); u32* largerPtr = smallerPtr + 1;
    u32 dummy32; const u32 windowLow = 1 + idxShift; u32 matchEndIdx = curr + 8 + 1;
    u32 bestLength = 8;
    u32 nbCompares = 1u << searchLog;
    hashT[h] = curr;
    while (nbCompares-- && matchIndex >= windowLow) {
      u32* const nextPtr = bt + 2 * (matchIndex & btMask);
      u32 ml = min(commonSmaller, commonLarger);
      const u32 m = matchIndex - 1 - idxShift;
      ml += count_eq(src, ipos + ml, m + ml, iend);
      if (ml > bestLength) { bestLength = ml; if (ml > matchEndIdx - matchIndex) matchEndIdx = matchIndex + ml; }
      if (ipos + ml == iend) break;
      if (src[m + ml] < src[ipos + ml]) {
        *smallerPtr = matchIndex; commonSmaller = ml;
        if (matchIndex <= btLow) { smallerPtr = &dummy32; break; }
        smallerPtr = nextPtr + 1; matchIndex = nextPtr[1];
      } else {
        *largerPtr = matchIndex; commonLarger = ml;
        if (matchIndex <= btLow) { largerPtr = &dummy32; break; }
        largerPtr = nextPtr; matchIndex = nextPtr[0];
      }
    }
    *smallerPtr = 0; *largerPtr = 0;
    u32 positions = 0;
    if (bestLength > 384) positions = min(192u, bestLength - 384);
    return max(positions, matchEndIdx - (curr + 8));
  }
  // ZSTD_BtGetAllMatches
  __device__ u32 get_all_matches(ZraOptMatch* matches, u32* nextToUpdate3, const u8* src, u32 ip, u32 iend, const u32* rep, u32 ll0, u32 lengthToBeat) {
    const u32 curr = ip + 1 + idxShift;
    if (curr < nextToUpdate) return 0;
    const u32 mlsH = tmpl_mls();
    for (u32 idx = nextToUpdate; idx < curr;) idx += insert_bt1(src, idx, iend, mlsH);
    nextToUpdate = curr;
    const u32 sufficient_len = min(targetLength, ZRA_OPT_NUM - 1);
    const u32 minMatch = mlsH == 3 ? 3u : 4u;
    const u32 h = hashN(src + ip, hashLog, mlsH);
    u32 matchIndex = hashT[h];
    const u32 btMask = (1u << (chainLog - 1)) - 1;
    u32 commonSmaller = 0, commonLarger = 0;
    const u32 dictLimit = 1 + idxShift;
    const u32 btLow = btMask >= curr ? 0 : curr - btMask;
    const u32 windowLow = lowest_at(curr, windowLog, idxShift), matchLow = windowLow ? windowLow : 1;
    u32* smallerPtr = bt + 2 * (curr & btMask); u32* largerPtr = smallerPtr + 1;
    u32 matchEndIdx = curr + 8 + 1, dummy32, mnum = 0;
    u32 nbCompares = 1u << searchLog;
    u32 bestLength = lengthToBeat - 1;
    {
      const u32 lastR = 3 + ll0;
      for (u32 repCode = ll0; repCode < lastR; repCode++) {
        const u32 repOffset = repCode == 3 ? rep[0] - 1 : rep[repCode];
        const u32 repIndex = curr - repOffset;
        u32 repLen = 0;
        if (repOffset - 1 < curr - dictLimit) {
          const bool same = minMatch == 3 ? ((ld32(src + ip) << 8) == (ld32(src + ip - repOffset) << 8)) : (ld32(src + ip) == ld32(src + ip - repOffset));
          if (repIndex >= windowLow && same) repLen = count_eq(src, ip + minMatch, ip + minMatch - repOffset, iend) + minMatch;
        }
        if (repLen > bestLength) {
          bestLength = repLen;
          matches[mnum].off = repCode - ll0; matches[mnum].len = repLen; mnum++;
          if (repLen > sufficient_len || ip + repLen == iend) return mnum;
        }
      }
    }
    if (mlsH == 3 && bestLength < 3) {
      u32 idx = *nextToUpdate3; const u32 target = curr;
      const u32 h3 = hash3of(src + ip, hashLog3);
      while (idx < target) { hash3[hash3of(src + (idx - 1 - idxShift), hashLog3)] = idx; idx++; }
      *nextToUpdate3 = target;
      const u32 matchIndex3 = hash3[h3];
      if (matchIndex3 >= matchLow && curr - matchIndex3 < (1u << 18)) {
        const u32 mlen = count_eq(src, ip, matchIndex3 - 1 - idxShift, iend);
        if (mlen >= 3) {
          bestLength = mlen;
          matches[0].off = (curr - matchIndex3) + 2; matches[0].len = mlen; mnum = 1;
          if (mlen > sufficient_len || ip + mlen == iend) { nextToUpdate = curr + 1; return 1; }
        }
      }
    }
    hashT[h] = curr;
    while (nbCompares-- && matchIndex >= matchLow) {
      u32* const nextPtr = bt + 2 * (matchIndex & btMask);
      u32 ml = min(commonSmaller, commonLarger);
      const u32 m = matchIndex - 1 - idxShift;
      ml += count_eq(src, ip + ml, m + ml, iend);
      if (ml > bestLength) {
        if (ml > matchEndIdx - matchIndex) matchEndIdx = matchIndex + ml;
        bestLength = ml;
        matches[mnum].off = (curr - matchIndex) + 2; matches[mnum].len = ml; mnum++;
        if (ml > ZRA_OPT_NUM || ip + ml == iend) break;
      }
      if (src[m + ml] < src[ip + ml]) {
        *smallerPtr = matchIndex; commonSmaller = ml;
        if (matchIndex <= btLow) { smallerPtr = &dummy32; break; }
        smallerPtr = nextPtr + 1; matchIndex = nextPtr[1];
      } else {
        *largerPtr = matchIndex; commonLarger = ml;
        if (matchIndex <= btLow) { largerPtr = &dummy32; break; }
        largerPtr = nextPtr; matchIndex = nextPtr[0];
      }
    }
    *smallerPtr = 0; *largerPtr = 0;
    nextToUpdate = matchEndIdx - 8;
    return mnum;
  }
  // ZSTD_compressBlock_opt_generic; `dry`: btultra2's statistics pass (nothing is emitted)
  __device__ u32 parse(const u8* src, u32 bs, u32 be, u32* rep, Emit& E, bool dry) {
    ZraOptCell* opt = o->table; ZraOptMatch* matches = o->matches;
    u32 anchor = bs, ip = bs;
    const u32 ilimit = be >= 8 ? be - 8 : 0;
    const u32 sufficient_len = min(targetLength, ZRA_OPT_NUM - 1);
    const u32 minMatch = minMatchParam == 3 ? 3u : 4u;
    u32 nextToUpdate3 = nextToUpdate;
    ZraOptCell lastSequence; lastSequence.price = 0; lastSequence.off = lastSequence.mlen = lastSequence.litlen = 0; lastSequence.rep[0] = lastSequence.rep[1] = lastSequence.rep[2] = 0;
    rescale_freqs(src + bs, be - bs);
    ip += (bs == 0) ? 1u : 0u;                          // ip += (ip == prefixStart)
    while (ip < ilimit) {
      u32 cur, last_pos = 0;
      bool shortcut = false;
      {
        const u32 litlen = ip - anchor, ll0 = !litlen;
        const u32 nb = get_all_matches(matches, &nextToUpdate3, src, ip, be, rep, ll0, minMatch);
        if (!nb) { ip++; continue; }
        for (int i = 0; i < 3; i++) opt[0].rep[i] = rep[i];
        opt[0].mlen = 0; opt[0].litlen = litlen;
        opt[0].price = (int)ll_price(litlen);
        const u32 maxML = matches[nb - 1].len, maxOffset = matches[nb - 1].off;
        if (maxML > sufficient_len) {
          lastSequence.litlen = litlen; lastSequence.mlen = maxML; lastSequence.off = maxOffset;
          cur = 0; last_pos = lastSequence.litlen + lastSequence.mlen;
          shortcut = true;
        } else {
          const u32 literalsPrice = (u32)opt[0].price + ll_price(0);
          u32 pos;
          for (pos = 1; pos < minMatch; pos++) opt[pos].price = 1 << 30;
          for (u32 k = 0; k < nb; k++) {
            const u32 offset = matches[k].off, end = matches[k].len;
            for (; pos <= end; pos++) {
              const u32 matchPrice = match_price(offset, pos);
              opt[pos].mlen = pos; opt[pos].off = offset; opt[pos].litlen = litlen; opt[pos].price = (int)(literalsPrice + matchPrice);
            }
          }
          last_pos = pos - 1;
        }
      }
      if (!shortcut) {
        for (cur = 1; cur <= last_pos; cur++) {
          const u32 inr = ip + cur;
          {
            const u32 litlen = opt[cur - 1].mlen == 0 ? opt[cur - 1].litlen + 1 : 1;
            const int price = opt[cur - 1].price + (int)raw_literals_cost(src + ip + cur - 1, 1) + (int)ll_price(litlen) - (int)ll_price(litlen - 1);
            if (price <= opt[cur].price) { opt[cur].mlen = 0; opt[cur].off = 0; opt[cur].litlen = litlen; opt[cur].price = price; }
          }
          if (opt[cur].mlen != 0) {
            const u32 prev = cur - opt[cur].mlen;
            u32 nr[3]; update_rep(nr, opt[prev].rep, opt[cur].off, opt[cur].litlen == 0);
            opt[cur].rep[0] = nr[0]; opt[cur].rep[1] = nr[1]; opt[cur].rep[2] = nr[2];
          } else { opt[cur].rep[0] = opt[cur - 1].rep[0]; opt[cur].rep[1] = opt[cur - 1].rep[1]; opt[cur].rep[2] = opt[cur - 1].rep[2]; }
          if (inr > ilimit) continue;
          if (cur == last_pos) break;
          if (lvl == 0 && opt[cur + 1].price <= opt[cur].price + 128) continue;
          {
            const u32 ll0 = opt[cur].mlen != 0;
            const u32 litlen = opt[cur].mlen == 0 ? opt[cur].litlen : 0;
            const u32 basePrice = (u32)opt[cur].price + ll_price(0);
            u32 crep[3] = {opt[cur].rep[0], opt[cur].rep[1], opt[cur].rep[2]};
            const u32 nb = get_all_matches(matches, &nextToUpdate3, src, inr, be, crep, ll0, minMatch);
            if (!nb) continue;
            const u32 maxML = matches[nb - 1].len;
            if (maxML > sufficient_len || cur + maxML >= ZRA_OPT_NUM) {
              lastSequence.mlen = maxML; lastSequence.off = matches[nb - 1].off; lastSequence.litlen = litlen;
              cur -= opt[cur].mlen == 0 ? opt[cur].litlen : 0;
              last_pos = cur + lastSequence.litlen + lastSequence.mlen;
              if (cur > ZRA_OPT_NUM) cur = 0;
              shortcut = true;
              break;
            }
            for (u32 k = 0; k < nb; k++) {
              const u32 offset = matches[k].off, lastML = matches[k].len, startML = k > 0 ? matches[k - 1].len + 1 : minMatch;
              for (u32 mlen = lastML; mlen >= startML; mlen--) {
                const u32 pos = cur + mlen;
                const int price = (int)(basePrice + match_price(offset, mlen));
                if (pos > last_pos || price < opt[pos].price) {
                  while (last_pos < pos) { opt[last_pos + 1].price = 1 << 30; last_pos++; }
                  opt[pos].mlen = mlen; opt[pos].off = offset; opt[pos].litlen = litlen; opt[pos].price = price;
                } else if (lvl == 0) break;
              }
            }
          }
        }
        if (!shortcut) {
          lastSequence = opt[last_pos];
          cur = last_pos > lastSequence.litlen + lastSequence.mlen ? last_pos - (lastSequence.litlen + lastSequence.mlen) : 0;
        }
      }
      // _shortestPath
      if (lastSequence.mlen != 0) { u32 nr[3]; update_rep(nr, opt[cur].rep, lastSequence.off, lastSequence.litlen == 0); rep[0] = nr[0]; rep[1] = nr[1]; rep[2] = nr[2]; }
      else { rep[0] = opt[cur].rep[0]; rep[1] = opt[cur].rep[1]; rep[2] = opt[cur].rep[2]; }
      {
        const u32 storeEnd = cur + 1; u32 storeStart = storeEnd, seqPos = cur;
        opt[storeEnd] = lastSequence;
        while (seqPos > 0) {
          const u32 backDist = opt[seqPos].litlen + opt[seqPos].mlen;
          storeStart--;
          opt[storeStart] = opt[seqPos];
          seqPos = seqPos > backDist ? seqPos - backDist : 0;
        }
        for (u32 sp = storeStart; sp <= storeEnd; sp++) {
          const u32 llen = opt[sp].litlen, mlen = opt[sp].mlen, offCode = opt[sp].off, advance = llen + mlen;
          if (mlen == 0) { ip = anchor + llen; continue; }
          update_stats(llen, src + anchor, offCode, mlen);
          if (!dry) E.put(llen, mlen, offCode + 1);
          anchor += advance; ip = anchor;
        }
        set_base_prices();
      }
    }
    return be - anchor;
  }
};
