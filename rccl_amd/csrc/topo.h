#pragma once
#include <vector>
// Hamiltonian ring set over the direct links of `adj` (n x n, 0/1); every ring starts at 0.
int raTopoRings(int n, const int* adj, int maxRings, std::vector<std::vector<int>>& rings);
// xGMI adjacency of a topology XML in the reference's format; 0 on success.
int raTopoParseXml(const char* text, int maxGpus, int* adj, int* ngpus);
// One xGMI hop between two GPUs, as hipExtGetLinkTypeAndHopCount reports it (HSA_AMD_LINK_INFO_TYPE_XGMI == 4): the
// links the reference turns into <xgmi> edges (src/graph/xml.cc:780-791).
int raTopoLinkIsDirect(unsigned type, unsigned hops);
