// Single-node topology: xGMI adjacency from a topology XML and ring-schedule synthesis.
//
// The reference reaches its ring set for an all-to-all xGMI node by matching the topology
// against pre-baked tables (src/graph/search.cc:970-986, src/graph/rome_models.cc:549-560:
// 7 forward Hamiltonian rings + their reverses, every directed link used twice).  Here the
// set is searched for directly: a decomposition of the directed link graph into
// Hamiltonian cycles, so that every link carries the same share of the traffic.
#include <string.h>

#include <string>
#include <vector>

#include "debug.h"
#include "rccl_amd.h"
#include "topo.h"

namespace {

struct Search {
  int n;
  int lambda;                       // times each directed link must be used
  std::vector<int> left;            // left[u*n+v]: remaining uses of link u->v
  std::vector<std::vector<int>> rings;
  int want;                         // rings to find
  long budget;

  bool extend(std::vector<int>& path, unsigned visited) {
    if (--budget < 0) return false;
    const int u = path.back();
    if ((int)path.size() == n) {
      if (left[u * n + path[0]] <= 0) return false;
      left[u * n + path[0]]--;
      rings.push_back(path);
      if ((int)rings.size() == want || nextRing()) return true;
      rings.pop_back();
      left[u * n + path[0]]++;
      return false;
    }
    // try the links with the most remaining uses first: keeps the leftover graph regular
    for (int pass = lambda; pass >= 1; pass--) {
      for (int v = 1; v < n; v++) {
        if ((visited >> v & 1) || left[u * n + v] != pass) continue;
        // a ring identical to an earlier one is allowed only when lambda > 1
        left[u * n + v]--;
        path.push_back(v);
        if (extend(path, visited | 1u << v)) return true;
        path.pop_back();
        left[u * n + v]++;
        if (budget < 0) return false;
      }
    }
    return false;
  }
  bool nextRing() {
    std::vector<int> path(1, 0);
    return extend(path, 1u);
  }
};

// All rings start at GPU 0.  Returns true when an exact cover with multiplicity lambda exists
// within the search budget.
bool exactCover(int n, const int* adj, int lambda, std::vector<std::vector<int>>& out) {
  int edges = 0;
  for (int u = 0; u < n; u++)
    for (int v = 0; v < n; v++)
      if (u != v && adj[u * n + v]) edges++;
  if (edges == 0 || (lambda * edges) % n) return false;
  Search s;
  s.n = n; s.lambda = lambda; s.want = lambda * edges / n; s.budget = 4000000;
  s.left.assign((size_t)n * n, 0);
  for (int u = 0; u < n; u++)
    for (int v = 0; v < n; v++)
      if (u != v && adj[u * n + v]) s.left[u * n + v] = lambda;
  if (!s.nextRing()) return false;
  out = s.rings;
  return true;
}

// Best effort when no exact cover is found: repeatedly take a Hamiltonian cycle over the
// least-used links.
bool greedyRing(int n, const int* adj, std::vector<int>& use, std::vector<int>& ring) {
  std::vector<int> path(1, 0);
  unsigned visited = 1;
  long budget = 200000;
  struct Rec {
    static bool go(int n, const int* adj, std::vector<int>& use, std::vector<int>& path, unsigned visited, long& budget) {
      if (--budget < 0) return false;
      const int u = path.back();
      if ((int)path.size() == n) return adj[u * n + path[0]] != 0;
      std::vector<int> cand;
      for (int v = 1; v < n; v++) if (!(visited >> v & 1) && adj[u * n + v]) cand.push_back(v);
      for (size_t i = 0; i < cand.size(); i++)
        for (size_t j = i + 1; j < cand.size(); j++)
          if (use[u * n + cand[j]] < use[u * n + cand[i]]) std::swap(cand[i], cand[j]);
      for (int v : cand) {
        path.push_back(v);
        if (go(n, adj, use, path, visited | 1u << v, budget)) return true;
        path.pop_back();
      }
      return false;
    }
  };
  if (!Rec::go(n, adj, use, path, visited, budget)) return false;
  for (int i = 0; i < n; i++) use[path[i] * n + path[(i + 1) % n]]++;
  ring = path;
  return true;
}

}  // namespace

int raTopoLinkIsDirect(unsigned type, unsigned hops) { return type == 4u && hops == 1u; }

int raTopoRings(int n, const int* adj, int maxRings, std::vector<std::vector<int>>& rings) {
  rings.clear();
  if (n < 1 || n > 16 || maxRings < 1) return 0;
  if (n == 1) { rings.push_back(std::vector<int>(1, 0)); return 1; }
  if (n == 2) {
    if (adj[1] && adj[2]) rings.push_back({0, 1});
    return (int)rings.size();
  }
  std::vector<std::vector<int>> found;
  bool symmetric = true;
  for (int u = 0; u < n; u++)
    for (int v = 0; v < n; v++)
      if ((adj[u * n + v] != 0) != (adj[v * n + u] != 0)) symmetric = false;
  if (exactCover(n, adj, 1, found)) {
    // forward set, then its mirror image: both directions of every link, twice the rings
    rings = found;
    if (symmetric)
      for (const auto& r : found) {
        std::vector<int> rev(r.rbegin(), r.rend());   // ends at 0: rotate so that it starts there
        std::vector<int> rot;
        rot.push_back(0);
        for (int i = 0; i + 1 < n; i++) rot.push_back(rev[i]);
        bool present = false;                          // a set already closed under reversal stays as it is
        for (const auto& f : found) present = present || f == rot;
        if (!present) rings.push_back(rot);
      }
  } else if (exactCover(n, adj, 2, found)) {
    rings = found;
  } else {
    std::vector<int> use((size_t)n * n, 0);
    for (int k = 0; k < n - 1; k++) {
      std::vector<int> r;
      if (!greedyRing(n, adj, use, r)) break;
      rings.push_back(r);
    }
  }
  if ((int)rings.size() > maxRings) rings.resize((size_t)maxRings);
  return (int)rings.size();
}

// ---------------------------------------------------------------------------------------
// Minimal reader for the reference's topology XML (written by src/graph/xml.cc; samples in
// tools/topo_expl/models).  Only what the xGMI adjacency needs: every <gpu> sits inside the
// <pci busid=...> of its device and lists <xgmi target=busid count=N/> children
// (src/graph/xml.cc:780-791).
namespace {
std::string attr(const std::string& tag, const char* name) {
  std::string key = std::string(" ") + name + "=\"";
  size_t p = tag.find(key);
  if (p == std::string::npos) return "";
  p += key.size();
  size_t q = tag.find('"', p);
  return q == std::string::npos ? "" : tag.substr(p, q - p);
}
}  // namespace

int raTopoParseXml(const char* text, int maxGpus, int* adj, int* ngpus) {
  struct Gpu { std::string busid; std::vector<std::string> targets; int rank; };
  std::vector<Gpu> gpus;
  std::vector<std::string> pciStack;
  int cur = -1;
  const std::string s(text);
  size_t i = 0;
  while ((i = s.find('<', i)) != std::string::npos) {
    size_t e = s.find('>', i);
    if (e == std::string::npos) break;
    std::string tag = s.substr(i + 1, e - i - 1);
    i = e + 1;
    if (tag.empty() || tag[0] == '?' || tag[0] == '!') continue;
    const bool closing = tag[0] == '/';
    const bool selfClosing = tag.back() == '/';
    std::string name = tag.substr(closing ? 1 : 0, tag.find_first_of(" /\t\n", closing ? 1 : 0) - (closing ? 1 : 0));
    if (closing) {
      if (name == "pci" && !pciStack.empty()) pciStack.pop_back();
      if (name == "gpu") cur = -1;
      continue;
    }
    if (name == "pci") {
      if (!selfClosing) pciStack.push_back(attr(tag, "busid"));
    } else if (name == "gpu") {
      Gpu g;
      g.busid = pciStack.empty() ? "" : pciStack.back();
      std::string r = attr(tag, "rank");
      g.rank = r.empty() ? (int)gpus.size() : atoi(r.c_str());
      gpus.push_back(g);
      cur = selfClosing ? -1 : (int)gpus.size() - 1;
    } else if (name == "xgmi" && cur >= 0) {
      gpus[(size_t)cur].targets.push_back(attr(tag, "target"));
    }
  }
  const int n = (int)gpus.size();
  *ngpus = n;
  if (n > maxGpus) return -1;
  memset(adj, 0, sizeof(int) * (size_t)maxGpus * maxGpus);
  for (int a = 0; a < n; a++)
    for (const auto& t : gpus[(size_t)a].targets)
      for (int b = 0; b < n; b++)
        if (b != a && gpus[(size_t)b].busid == t) adj[a * n + b] = 1;
  return 0;
}

extern "C" __attribute__((visibility("default"))) ncclResult_t rcclAmdTopoRings(int ngpus, const int* adj, int maxRings, int* rings, int* nrings) {
  if (!adj || !rings || !nrings || ngpus < 1 || ngpus > 16 || maxRings < 1) return ncclInvalidArgument;
  std::vector<std::vector<int>> r;
  const int k = raTopoRings(ngpus, adj, maxRings, r);
  for (int i = 0; i < k; i++)
    for (int j = 0; j < ngpus; j++) rings[i * ngpus + j] = r[(size_t)i][(size_t)j];
  *nrings = k;
  return ncclSuccess;
}

extern "C" __attribute__((visibility("default"))) ncclResult_t rcclAmdTopoParseXml(const char* xmlText, int maxGpus, int* adj, int* ngpus) {
  if (!xmlText || !adj || !ngpus || maxGpus < 1) return ncclInvalidArgument;
  return raTopoParseXml(xmlText, maxGpus, adj, ngpus) == 0 ? ncclSuccess : ncclInvalidArgument;
}

// Adjacency and default schedule from a link table (linkType / hops: n x n as hipExtGetLinkTypeAndHopCount reports them,
// diagonal ignored): adj[a][b] = 1 when both directions of the pair are one xGMI hop; *allDirect = every pair is.
extern "C" __attribute__((visibility("default"))) ncclResult_t rcclAmdTopoFromLinks(int ngpus, const unsigned* linkType, const unsigned* hops,
                                                                                   int* adj, int* allDirect) {
  if (ngpus < 1 || ngpus > 16 || !linkType || !hops || !adj) return ncclInvalidArgument;
  int all = 1;
  for (int a = 0; a < ngpus; a++)
    for (int b = 0; b < ngpus; b++) {
      const int d = a != b && raTopoLinkIsDirect(linkType[a * ngpus + b], hops[a * ngpus + b]) &&
                    raTopoLinkIsDirect(linkType[b * ngpus + a], hops[b * ngpus + a]);
      adj[a * ngpus + b] = d;
      if (a != b && !d) all = 0;
    }
  if (allDirect) *allDirect = all;
  return ncclSuccess;
}
