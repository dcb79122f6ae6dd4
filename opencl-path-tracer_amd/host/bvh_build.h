// Bottom-level BVH builders (CPU).  Input producer for the device `intersect` kernels; same
// contract as the reference's src/bvh/bvh_build.h:10-15: returns (root, triangles re-emitted in
// leaf order -- duplicated by spatial splits --, nodes) where leaves address a contiguous triangle
// range and an inner node's children are the adjacent pair (left, left+1).  Node 0 is the root and
// node 1 is an unused pad so that every sibling pair is 2-aligned.
//
// Constants follow the reference (src/bvh/bvh_build.cpp:15-18, bvh_object_split.cpp:10,
// bvh_spatial_split.cpp:10-12): leaf <= 3 primitives, Ct = 1.5, Ci = 1.0, 32 object bins,
// 8 spatial bins, alpha = 1e-5, reference unsplitting on.  Deliberate differences, documented in
// binned-SAH tries axes {0,1,2} (the reference passes {1,2,3}); the
// spatial split competes with the object split on SAH (SBVH paper) instead of only against the
// leaf cost; a depth cap with median fallback bounds the traversal stack.
#pragma once
#include "../../include/ptamd.h"
#include "aabb.h"
#include <cstdint>
#include <vector>

namespace raytracer {

using VertexSceneData = pt_vertex;
using TriangleSceneData = pt_triangle;
using SubBVHNode = pt_sub_bvh_node;
using TopBVHNode = pt_top_bvh_node;

enum class BvhBuilder : int { BinnedSAH = 0, BinnedFast = 1, SpatialSplit = 2 };

constexpr int kMaxBvhDepth = 60; // device traversal stack: 32 LDS entries + 32 spill entries

struct BvhBuildResult {
    uint32_t rootNode = 0;
    std::vector<TriangleSceneData> triangles;
    std::vector<SubBVHNode> nodes;
    std::vector<uint32_t> originalTriangle; // triangles[i] is input triangle originalTriangle[i]
};

BvhBuildResult buildBVH(const VertexSceneData* vertices, size_t numVertices, const TriangleSceneData* triangles, size_t numTriangles, BvhBuilder kind);

// reference entry-point names
inline BvhBuildResult buildBinnedBVH(const std::vector<VertexSceneData>& v, const std::vector<TriangleSceneData>& t) { return buildBVH(v.data(), v.size(), t.data(), t.size(), BvhBuilder::BinnedSAH); }
inline BvhBuildResult buildBinnedFastBVH(const std::vector<VertexSceneData>& v, const std::vector<TriangleSceneData>& t) { return buildBVH(v.data(), v.size(), t.data(), t.size(), BvhBuilder::BinnedFast); }
inline BvhBuildResult buildSpatialSplitBVH(const std::vector<VertexSceneData>& v, const std::vector<TriangleSceneData>& t) { return buildBVH(v.data(), v.size(), t.data(), t.size(), BvhBuilder::SpatialSplit); }

// Bottom-up box refit for an unchanged topology (reference: src/bvh/refit_bvh.cpp:6-34).
void refitBVH(std::vector<SubBVHNode>& nodes, uint32_t root, const std::vector<TriangleSceneData>& triangles, const std::vector<VertexSceneData>& vertices);

// Structural self-check (reference: BvhTester, src/bvh/bvh_test.cpp:23-139).
struct BvhStats {
    uint32_t numNodes = 0, numLeaves = 0, numTriangleRefs = 0, maxDepth = 0, maxLeafSize = 0;
    bool childrenInsideParents = true; // every child box within its parent box
    bool trianglesInsideLeaves = true; // every leaf triangle overlaps... (exact containment for object splits)
    bool allTrianglesReferenced = true;
};
BvhStats checkBVH(const BvhBuildResult& bvh, const VertexSceneData* vertices, size_t numInputTriangles, bool exactLeafContainment);

} // namespace raytracer
