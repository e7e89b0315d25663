// Test infrastructure: mutation fuzzer for the WAV / FLAC readers of espresso_amd/csrc/ingest.hip, built by tests/test_ingest.py with
// g++ -fsanitize=address,undefined (the reader source is plain host C++ and is included here as it lies).  Every mutated file goes
// through ea_audio_probe / ea_audio_read_i16 / ea_audio_verify / ea_audio_read_batch_i16: whatever the bytes are, the readers must
// return (a sample count or an error code) — any out-of-bounds access, signed overflow UB or leak aborts the process.
//   fuzz_ingest <dir with seed files> <iterations per seed> <seed>
#include "../espresso_amd/csrc/ingest.hip"

#include <dirent.h>
#include <random>
#include <string>

static std::vector<uint8_t> slurp(const std::string& p) {
  std::vector<uint8_t> v;
  FILE* f = fopen(p.c_str(), "rb");
  if (!f) return v;
  uint8_t buf[65536];
  size_t n;
  while ((n = fread(buf, 1, sizeof(buf), f)) > 0) v.insert(v.end(), buf, buf + n);
  fclose(f);
  return v;
}

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  const std::string dir = argv[1];
  const int iters = atoi(argv[2]);
  std::mt19937_64 rng((uint64_t)atoll(argv[3]));
  std::vector<std::string> seeds;
  if (DIR* d = opendir(dir.c_str())) {
    while (dirent* e = readdir(d)) {
      const std::string n = e->d_name;
      if (n.size() > 4 && n.compare(0, 5, "seed_") == 0) seeds.push_back(dir + "/" + n);
    }
    closedir(d);
  }
  if (seeds.empty()) return 3;
  const std::string tmp = dir + "/mutant.bin", tmp2 = dir + "/mutant2.bin";
  long calls = 0, decoded = 0, errors = 0;
  for (const std::string& s : seeds) {
    const std::vector<uint8_t> orig = slurp(s);
    if (orig.empty()) return 4;
    for (int it = 0; it < iters; ++it) {
      std::vector<uint8_t> m = orig;
      const int kind = (int)(rng() % 6);
      if (kind == 0) {  // flip a few bytes anywhere
        for (int k = 0, n = 1 + (int)(rng() % 8); k < n; ++k) m[rng() % m.size()] ^= (uint8_t)(1u << (rng() % 8));
      } else if (kind == 1) {  // garbage in the header region
        for (int k = 0, n = 1 + (int)(rng() % 16); k < n; ++k) m[rng() % std::min<size_t>(m.size(), 96)] = (uint8_t)rng();
      } else if (kind == 2) {  // truncate
        m.resize((size_t)(rng() % (m.size() + 1)));
      } else if (kind == 3) {  // overwrite a run with one value (0x00 / 0xff / random)
        const size_t a = rng() % m.size(), len = 1 + rng() % 64;
        const uint8_t v = (rng() & 1) ? 0xff : ((rng() & 1) ? 0x00 : (uint8_t)rng());
        for (size_t k = a; k < std::min(m.size(), a + len); ++k) m[k] = v;
      } else if (kind == 4) {  // duplicate a slice somewhere else (sizes in headers no longer match)
        const size_t a = rng() % m.size(), len = 1 + rng() % 256, b = rng() % m.size();
        std::vector<uint8_t> piece(m.begin() + a, m.begin() + std::min(m.size(), a + len));
        m.insert(m.begin() + b, piece.begin(), piece.end());
      } else {  // huge numbers in 4-byte fields
        const size_t a = rng() % m.size();
        for (size_t k = a; k < std::min(m.size(), a + 4); ++k) m[k] = 0xff;
      }
      FILE* f = fopen(tmp.c_str(), "wb");
      if (!f) return 5;
      if (!m.empty()) fwrite(m.data(), 1, m.size(), f);
      fclose(f);
      long n = 0;
      int sr = 0, ch = 0, bits = 0;
      const int rc = ea_audio_probe(tmp.c_str(), &n, &sr, &ch, &bits);
      ++calls;
      // capacity from the (untrusted) header, capped; also a deliberately short buffer
      const long cap = rc == 0 ? std::min<long>(std::max<long>(n, 0), 1L << 22) : 4096;
      std::vector<int16_t> out((size_t)cap + 1);
      const long got = ea_audio_read_i16(tmp.c_str(), out.data(), cap, &sr);
      if (got >= 0) ++decoded; else ++errors;
      if (got > cap) return 6;  // wrote past what it was given
      std::vector<int16_t> small(17);
      (void)ea_audio_read_i16(tmp.c_str(), small.data(), 16, &sr);
      (void)ea_audio_verify(tmp.c_str());
      if ((it & 15) == 0) {  // the threaded batch entry: the mutant next to its seed
        const char* paths[3] = {s.c_str(), tmp.c_str(), tmp2.c_str()};  // (tmp2 does not exist)
        std::vector<int16_t> dst((size_t)(1 << 20));
        const long offs[4] = {0, 300000, 600000, 700000};
        long lens[3];
        int rates[3];
        (void)ea_audio_read_batch_i16(paths, 3, dst.data(), offs, 3, lens, rates);
        for (int i = 0; i < 3; ++i)
          if (lens[i] > offs[i + 1] - offs[i]) return 7;
      }
    }
  }
  printf("fuzz_ingest: %zu seeds, %ld mutants, %ld decoded, %ld rejected\n", seeds.size(), calls, decoded, errors);
  return 0;
}
