#pragma once
#include <vector>
// Hamiltonian ring set over the direct links of `adj` (n x n, 0/1); every ring starts at 0.
int raTopoRings(int n, const int* adj, int maxRings, std::vector<std::vector<int>>& rings);
// xGMI adjacency of a topology XML in the reference's format; 0 on success.
int raTopoParseXml(const char* text, int maxGpus, int* adj, int* ngpus);
