#include "mesh.h"
#include "parallel.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <tuple>
#include <sstream>
#include <stdexcept>

namespace raytracer {

Mesh::Mesh(const float* positions, const float* normals, const float* texCoords, size_t numVertices,
    const uint32_t* indices, const uint32_t* materialIndex, size_t numTriangles,
    const std::vector<Material>& materials, BvhBuilder builder, const std::string& bvhCacheFile)
    : m_materials(materials), m_builder(builder)
{
    if (materials.empty())
        throw std::invalid_argument("Mesh: at least one material is required");
    const auto t0 = std::chrono::steady_clock::now();
    m_vertices.resize(numVertices);
    std::memset(m_vertices.data(), 0, numVertices * sizeof(VertexSceneData));
    for (size_t i = 0; i < numVertices; i++) {
        for (int k = 0; k < 3; k++)
            m_vertices[i].vertex[k] = positions[3 * i + k];
        m_vertices[i].vertex[3] = 1.0f; // homogeneous w, as the reference's glm::vec4 positions (mesh.cpp:78)
        if (normals)
            for (int k = 0; k < 3; k++)
                m_vertices[i].normal[k] = normals[3 * i + k];
        if (texCoords) {
            m_vertices[i].texCoord[0] = texCoords[2 * i];
            m_vertices[i].texCoord[1] = texCoords[2 * i + 1];
        }
        m_bounds.fit(vec3(positions[3 * i], positions[3 * i + 1], positions[3 * i + 2]));
    }
    m_inputTriangles.resize(numTriangles);
    for (size_t t = 0; t < numTriangles; t++) {
        for (int k = 0; k < 3; k++) {
            if (indices[3 * t + k] >= numVertices) // the smooth-normal pass and the builders index m_vertices with these
                throw std::invalid_argument("Mesh: vertex index out of range");
            m_inputTriangles[t].indices[k] = indices[3 * t + k];
        }
        uint32_t mi = materialIndex ? materialIndex[t] : 0;
        if (mi >= materials.size())
            throw std::invalid_argument("Mesh: material index out of range");
        m_inputTriangles[t].materialIndex = mi;
    }
    const auto t1 = std::chrono::steady_clock::now();
    if (!normals)
        generateSmoothNormals();
    const auto t2 = std::chrono::steady_clock::now();
    if (bvhCacheFile.empty() || !loadBvh(bvhCacheFile)) {
        m_bvh = buildBVH(m_vertices.data(), m_vertices.size(), m_inputTriangles.data(), m_inputTriangles.size(), builder);
        if (!bvhCacheFile.empty())
            storeBvh(bvhCacheFile);
    }
    const auto t3 = std::chrono::steady_clock::now();
    // emissive triangles are listed once per INPUT triangle (spatial splits may duplicate references)
    std::vector<uint8_t> listed(numTriangles, 0);
    for (size_t i = 0; i < m_bvh.triangles.size(); i++) {
        uint32_t orig = m_bvh.originalTriangle[i];
        if (m_materials[m_bvh.triangles[i].materialIndex].type == PT_MAT_EMISSIVE && !listed[orig]) {
            listed[orig] = 1;
            m_emissive.push_back((uint32_t)i);
        }
    }
    if (std::getenv("PTAMD_BUILD_TIMING")) {
        const auto t4 = std::chrono::steady_clock::now();
        auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "[ptamd_host] Mesh: arrays %.3f ms, smooth normals %.3f, tree %.3f, emissive list %.3f\n", ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, t4));
    }
}

static const uint32_t BVH_FILE_FORMAT_VERSION = 1; // mesh.cpp:200

void Mesh::storeBvh(const std::string& fileName) const
{
    std::ofstream out(fileName, std::ios::out | std::ios::binary | std::ios::trunc);
    if (!out)
        throw std::runtime_error("cannot write " + fileName);
    refittedBvh(); // (the boxes of a refitted mesh are made when asked for)
    const uint32_t root = m_bvh.rootNode, numNodes = (uint32_t)m_bvh.nodes.size(), numTriangles = (uint32_t)m_bvh.triangles.size();
    out.write((const char*)&BVH_FILE_FORMAT_VERSION, 4);
    out.write((const char*)&root, 4);
    out.write((const char*)&numNodes, 4);
    out.write((const char*)m_bvh.nodes.data(), (std::streamsize)numNodes * sizeof(SubBVHNode));
    out.write((const char*)&numTriangles, 4);
    out.write((const char*)m_bvh.triangles.data(), (std::streamsize)numTriangles * sizeof(TriangleSceneData));
    out << std::endl; // the reference ends the file with one (mesh.cpp:223)
    if (!out)
        throw std::runtime_error("short write to " + fileName);
}

bool Mesh::loadBvh(const std::string& fileName)
{
    std::ifstream in(fileName, std::ios::binary);
    if (!in)
        return false;
    in.seekg(0, std::ios::end);
    const uint64_t fileSize = (uint64_t)in.tellg();
    in.seekg(0);
    uint32_t version = 0, root = 0, numNodes = 0, numTriangles = 0;
    in.read((char*)&version, 4);
    in.read((char*)&root, 4);
    in.read((char*)&numNodes, 4);
    if (!in || version != BVH_FILE_FORMAT_VERSION || numNodes == 0 || root >= numNodes || 12 + (uint64_t)numNodes * sizeof(SubBVHNode) + 4 > fileSize)
        return false;
    BvhBuildResult r;
    r.rootNode = root;
    r.nodes.resize(numNodes);
    in.read((char*)r.nodes.data(), (std::streamsize)numNodes * sizeof(SubBVHNode));
    in.read((char*)&numTriangles, 4);
    const uint64_t payload = 16 + (uint64_t)numNodes * sizeof(SubBVHNode) + (uint64_t)numTriangles * sizeof(TriangleSceneData);
    if (!in || numTriangles == 0 || (payload != fileSize && payload + 1 != fileSize)) // with or without the trailing newline
        return false;
    r.triangles.resize(numTriangles);
    in.read((char*)r.triangles.data(), (std::streamsize)numTriangles * sizeof(TriangleSceneData));
    if (!in)
        return false;
    // the file must describe THIS mesh: every reference is one of the input triangles, every input triangle is there
    struct Key {
        uint32_t a, b, c, m;
        bool operator<(const Key& o) const { return std::tie(a, b, c, m) < std::tie(o.a, o.b, o.c, o.m); }
    };
    std::map<Key, uint32_t> inputOf;
    for (size_t t = 0; t < m_inputTriangles.size(); t++) {
        const auto& tri = m_inputTriangles[t];
        inputOf.emplace(Key { tri.indices[0], tri.indices[1], tri.indices[2], tri.materialIndex }, (uint32_t)t);
    }
    std::vector<uint8_t> seen(m_inputTriangles.size(), 0);
    r.originalTriangle.resize(numTriangles);
    for (uint32_t i = 0; i < numTriangles; i++) {
        const auto& tri = r.triangles[i];
        auto it = inputOf.find(Key { tri.indices[0], tri.indices[1], tri.indices[2], tri.materialIndex });
        if (it == inputOf.end())
            return false;
        r.originalTriangle[i] = it->second;
        seen[it->second] = 1;
    }
    for (const auto& tri : m_inputTriangles) // duplicates of one input triangle share a key: look each one up
        if (!seen[inputOf.at(Key { tri.indices[0], tri.indices[1], tri.indices[2], tri.materialIndex })])
            return false;
    // structure first (checkBVH assumes a well-formed tree): children lie after their parent, so there are no
    // cycles, and every index is in range
    {
        std::vector<uint32_t> todo { root };
        while (!todo.empty()) {
            const uint32_t i = todo.back();
            todo.pop_back();
            const SubBVHNode& n = r.nodes[i];
            if (n.triangleCount != 0) {
                if ((uint64_t)n.leftChildOrFirstTriangle + n.triangleCount > numTriangles)
                    return false;
            } else {
                const uint32_t l = n.leftChildOrFirstTriangle;
                if (l <= i || (uint64_t)l + 1 >= numNodes)
                    return false;
                todo.push_back(l);
                todo.push_back(l + 1);
            }
        }
    }
    // an object-split tree holds whole triangles in its leaves; only a spatial-split tree may hold clipped references (checkBVH then asks
    // that the leaves referencing a triangle cover its EXTENT between them -- a heuristic, necessary but not sufficient: a doctored file that
    // shrinks an interior clipped piece leaves a hole the extents do not show; a cache that must be trusted needs a content hash)
    const BvhStats st = checkBVH(r, m_vertices.data(), m_inputTriangles.size(), m_builder != BvhBuilder::SpatialSplit);
    if (!st.childrenInsideParents || !st.trianglesInsideLeaves || !st.allTrianglesReferenced)
        return false;
    m_bvh = std::move(r);
    m_bvhFromCache = true;
    return true;
}


namespace {
struct MtlEntry {
    vec3 kd { 0.6f, 0.6f, 0.6f }; // Assimp's default diffuse colour
    vec3 ke { 0.0f, 0.0f, 0.0f };
    std::string mapKd; // diffuse texture, as written in the file
};

std::map<std::string, MtlEntry> readMtl(const std::string& path)
{
    std::map<std::string, MtlEntry> out;
    std::ifstream in(path);
    std::string line, cur;
    while (std::getline(in, line)) {
        std::istringstream ls(line);
        std::string tag;
        if (!(ls >> tag) || tag[0] == '#')
            continue;
        if (tag == "newmtl") {
            ls >> cur;
            out[cur] = MtlEntry {};
        } else if (!cur.empty() && (tag == "Kd" || tag == "Ke")) {
            vec3 c;
            if (ls >> c.x >> c.y >> c.z)
                (tag == "Kd" ? out[cur].kd : out[cur].ke) = c;
        } else if (!cur.empty() && tag == "map_Kd") {
            std::string rest, word; // options (-s, -o, -bm ...) come first, the file name last
            while (ls >> word)
                rest = word;
            for (char& ch : rest)
                if (ch == '\\')
                    ch = '/';
            out[cur].mapKd = rest;
        }
    }
    return out;
}
} // namespace

std::shared_ptr<Mesh> Mesh::fromOBJ(const std::string& path, const Material* overrideMaterial, const Transform& offset, BvhBuilder builder,
    const std::string& bvhCacheFile, UniqueTextureFiles* textures)
{
    std::ifstream in(path);
    if (!in)
        throw std::runtime_error("cannot open " + path);
    const std::string folder = path.find_last_of("/\\") == std::string::npos ? std::string() : path.substr(0, path.find_last_of("/\\") + 1);
    std::vector<vec3> P, N;
    std::vector<float> UV; // 2 per entry
    std::map<std::string, MtlEntry> mtl;
    std::vector<Material> materials;
    std::map<std::string, uint32_t> materialOf;
    uint32_t curMaterial = 0;
    bool haveMaterial = false;
    auto useMaterial = [&](const std::string& name) {
        if (overrideMaterial) {
            if (materials.empty())
                materials.push_back(*overrideMaterial);
            curMaterial = 0;
        } else {
            auto it = materialOf.find(name);
            if (it == materialOf.end()) {
                const MtlEntry e = mtl.count(name) ? mtl[name] : MtlEntry {};
                const bool emissive = e.ke.x != 0.0f || e.ke.y != 0.0f || e.ke.z != 0.0f; // mesh.cpp:58
                if (emissive)
                    materials.push_back(Material::Emissive(e.ke));
                else if (textures && !e.mapKd.empty()) // mesh.cpp:61-66
                    materials.push_back(Material::Diffuse(textures->add(folder + e.mapKd), e.kd));
                else
                    materials.push_back(Material::Diffuse(e.kd));
                it = materialOf.emplace(name, (uint32_t)materials.size() - 1).first;
            }
            curMaterial = it->second;
        }
        haveMaterial = true;
    };
    struct Corner {
        int v, t, n;
        bool operator<(const Corner& o) const { return std::tie(v, t, n) < std::tie(o.v, o.t, o.n); }
    };
    std::map<Corner, uint32_t> welded;
    std::vector<Corner> corners; // unique corners in first-use order
    std::vector<uint32_t> indices, materialIndex;
    bool allNormals = true;
    std::string line;
    size_t lineNo = 0;
    while (std::getline(in, line)) {
        lineNo++;
        if (!line.empty() && line.back() == '\r')
            line.pop_back();
        std::istringstream ls(line);
        std::string tag;
        if (!(ls >> tag) || tag[0] == '#')
            continue;
        if (tag == "v") {
            vec3 p;
            if (!(ls >> p.x >> p.y >> p.z))
                throw std::runtime_error(path + ":" + std::to_string(lineNo) + ": bad vertex");
            P.push_back(p);
        } else if (tag == "vn") {
            vec3 n;
            if (!(ls >> n.x >> n.y >> n.z))
                throw std::runtime_error(path + ":" + std::to_string(lineNo) + ": bad normal");
            N.push_back(n);
        } else if (tag == "vt") {
            float u = 0.0f, v = 0.0f;
            if (!(ls >> u))
                throw std::runtime_error(path + ":" + std::to_string(lineNo) + ": bad texture coordinate");
            ls >> v;
            UV.push_back(u), UV.push_back(v);
        } else if (tag == "mtllib") {
            std::string name;
            ls >> name;
            for (auto& kv : readMtl(folder + name))
                mtl[kv.first] = kv.second;
        } else if (tag == "usemtl") {
            std::string name;
            ls >> name;
            useMaterial(name);
        } else if (tag == "f") {
            if (!haveMaterial)
                useMaterial(""); // faces before any usemtl: the default material
            std::vector<uint32_t> poly;
            std::string tok;
            while (ls >> tok) {
                Corner c { 0, 0, 0 };
                // v, v/t, v//n, v/t/n ; 1-based, negative = relative to the end
                int* fields[3] = { &c.v, &c.t, &c.n };
                size_t pos = 0;
                for (int k = 0; k < 3 && pos <= tok.size(); k++) {
                    size_t slash = tok.find('/', pos);
                    std::string part = tok.substr(pos, slash == std::string::npos ? std::string::npos : slash - pos);
                    if (!part.empty())
                        *fields[k] = std::stoi(part);
                    if (slash == std::string::npos)
                        break;
                    pos = slash + 1;
                }
                auto resolve = [&](int idx, size_t count, const char* what) -> int {
                    if (idx == 0)
                        return -1;
                    const long long r = idx > 0 ? idx - 1 : (long long)count + idx;
                    if (r < 0 || r >= (long long)count)
                        throw std::runtime_error(path + ":" + std::to_string(lineNo) + ": " + what + " index out of range");
                    return (int)r;
                };
                c.v = resolve(c.v, P.size(), "vertex");
                c.t = resolve(c.t, UV.size() / 2, "texture coordinate");
                c.n = resolve(c.n, N.size(), "normal");
                if (c.v < 0)
                    throw std::runtime_error(path + ":" + std::to_string(lineNo) + ": face corner without a vertex");
                if (c.n < 0)
                    allNormals = false;
                auto it = welded.find(c);
                if (it == welded.end()) {
                    it = welded.emplace(c, (uint32_t)corners.size()).first;
                    corners.push_back(c);
                }
                poly.push_back(it->second);
            }
            for (size_t k = 2; k < poly.size(); k++) { // fan triangulation; points and lines are dropped (mesh.cpp:108-110)
                indices.push_back(poly[0]), indices.push_back(poly[k - 1]), indices.push_back(poly[k]);
                materialIndex.push_back(curMaterial);
            }
        }
    }
    if (indices.empty())
        throw std::runtime_error(path + ": no faces");
    const mat4 M = offset.matrix();
    const mat4 Minv = inverse(M);
    std::vector<float> positions(corners.size() * 3), normals, uvs(corners.size() * 2, 0.0f);
    if (allNormals)
        normals.resize(corners.size() * 3);
    for (size_t i = 0; i < corners.size(); i++) {
        const vec3 p = (M * vec4(P[corners[i].v], 1.0f)).xyz();
        positions[3 * i] = p.x, positions[3 * i + 1] = p.y, positions[3 * i + 2] = p.z;
        if (allNormals) { // normal matrix = transpose(inverse(M)) (mesh_helpers.cpp:20-23), w = 0
            const vec3 n = N[corners[i].n];
            for (int r = 0; r < 3; r++)
                normals[3 * i + r] = Minv.m[r][0] * n.x + Minv.m[r][1] * n.y + Minv.m[r][2] * n.z;
        }
        if (corners[i].t >= 0)
            uvs[2 * i] = UV[2 * corners[i].t], uvs[2 * i + 1] = UV[2 * corners[i].t + 1];
    }
    return std::make_shared<Mesh>(positions.data(), allNormals ? normals.data() : nullptr, uvs.data(), corners.size(), indices.data(),
        materialIndex.data(), indices.size() / 3, materials, builder, bvhCacheFile);
}

void Mesh::generateSmoothNormals()
{
    // area-weighted smooth normals (cross product length = 2*area): normal(v) = normalize(sum of the face normals of the triangles at v).
    // Two passes, each over disjoint ranges on the library's worker threads: face normals, then per vertex the sum over its incident
    // triangles in ascending triangle order -- the order a scatter over the triangles adds them in, so the result does not depend on
    // the number of threads (and is what the single-threaded scatter of rounds 1-4 produced).
    const size_t numVertices = m_vertices.size(), numTriangles = m_inputTriangles.size();
    if (m_cornerStart.size() != numVertices + 1) { // the topology never changes: built once
        m_cornerStart.assign(numVertices + 1, 0);
        for (const auto& tri : m_inputTriangles)
            for (int k = 0; k < 3; k++)
                m_cornerStart[tri.indices[k] + 1]++;
        for (size_t v = 0; v < numVertices; v++)
            m_cornerStart[v + 1] += m_cornerStart[v];
        m_cornerTri.resize(numTriangles * 3);
        std::vector<uint32_t> fill(m_cornerStart.begin(), m_cornerStart.end() - 1);
        for (size_t t = 0; t < numTriangles; t++)
            for (int k = 0; k < 3; k++)
                m_cornerTri[fill[m_inputTriangles[t].indices[k]]++] = (uint32_t)t;
    }
    m_faceNormal.resize(numTriangles);
    WorkerPool& pool = WorkerPool::get();
    pool.parallelFor(numTriangles, 1024, [&](size_t begin, size_t end) {
        for (size_t t = begin; t < end; t++) {
            const auto& tri = m_inputTriangles[t];
            auto P = [&](int k) { const float* p = m_vertices[tri.indices[k]].vertex; return vec3(p[0], p[1], p[2]); };
            m_faceNormal[t] = cross(P(1) - P(0), P(2) - P(0));
        }
    });
    pool.parallelFor(numVertices, 1024, [&](size_t begin, size_t end) {
        for (size_t i = begin; i < end; i++) {
            vec3 acc;
            for (uint32_t c = m_cornerStart[i]; c < m_cornerStart[i + 1]; c++)
                acc += m_faceNormal[m_cornerTri[c]];
            const float len = length(acc);
            const vec3 n = len > 0.0f ? acc / len : vec3(0, 1, 0);
            m_vertices[i].normal[0] = n.x, m_vertices[i].normal[1] = n.y, m_vertices[i].normal[2] = n.z;
        }
    });
}

const BvhBuildResult& Mesh::refittedBvh() const
{
    if (m_boxesStale) {
        refitBVH(m_bvh.nodes, m_bvh.rootNode, m_bvh.triangles, m_vertices);
        m_boxesStale = false;
    }
    return m_bvh;
}

// A deformed frame of the same mesh (MeshSequence with m_refitting, reference src/model/mesh_sequence.cpp:81-97): new positions
// (and normals) for the same vertices and triangles; the BVH keeps its topology and leaf order, its boxes are recomputed bottom-up
// (refitBVH, src/bvh/refit_bvh.cpp:6-34).  Valid for any builder -- boxes of spatial-split references then cover the whole
// triangle, looser than the clipped ones but conservative.
void Mesh::refit(const float* positions, const float* normals)
{
    m_bounds = AABB();
    for (size_t i = 0; i < m_vertices.size(); i++) {
        for (int k = 0; k < 3; k++)
            m_vertices[i].vertex[k] = positions[3 * i + k];
        if (normals)
            for (int k = 0; k < 3; k++)
                m_vertices[i].normal[k] = normals[3 * i + k];
        m_bounds.fit(vec3(positions[3 * i], positions[3 * i + 1], positions[3 * i + 2]));
    }
    if (!normals)
        generateSmoothNormals();
    m_boxesStale = true; // refitBVH runs when the nodes are asked for (refittedBvh); the device library refits its own copy (pt_refit_vertices)
    m_bvhFromCache = false;
    m_generation++;
}

std::shared_ptr<Mesh> Mesh::fromPLY(const std::string& path, const Material& material, BvhBuilder builder)
{
    std::ifstream in(path, std::ios::binary);
    if (!in)
        throw std::runtime_error("cannot open " + path);
    std::string line;
    size_t nv = 0, nf = 0;
    bool ascii = true;
    int vertexProps = 0;
    bool inVertex = false;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r')
            line.pop_back();
        std::istringstream ls(line);
        std::string tok;
        ls >> tok;
        if (tok == "format") {
            ls >> tok;
            ascii = (tok == "ascii");
            if (!ascii && tok != "binary_little_endian")
                throw std::runtime_error("PLY: unsupported format " + tok);
        } else if (tok == "element") {
            std::string what;
            ls >> what;
            inVertex = (what == "vertex");
            if (what == "vertex") ls >> nv;
            if (what == "face") ls >> nf;
        } else if (tok == "property" && inVertex) {
            vertexProps++;
        } else if (tok == "end_header") {
            break;
        }
    }
    if (vertexProps < 3)
        throw std::runtime_error("PLY: vertices need x y z");
    if (vertexProps > 64)
        throw std::runtime_error("PLY: more than 64 vertex properties");
    // The counts of the header are untrusted: held against what the file can hold BEFORE anything is sized by them (a header that
    // announces four billion vertices must cost an error message, not 48 GB and four billion reads of an exhausted stream -- found by
    // tools/fuzz_loaders.py), and every read is checked: a truncated body is an error, not a mesh of zeros.
    const std::streamoff bodyStart = in.tellg();
    in.seekg(0, std::ios::end);
    const uint64_t bodyBytes = bodyStart < 0 ? 0 : (uint64_t)(in.tellg() - bodyStart);
    in.seekg(bodyStart);
    const uint64_t minVertex = ascii ? 6 : (uint64_t)vertexProps * sizeof(float), minFace = ascii ? 8 : 1 + 3 * sizeof(int32_t); // "0 0 0\n", "3 0 0 0\n"
    if (nv == 0 || nf == 0 || nv > bodyBytes / minVertex || nf > bodyBytes / minFace || nv * minVertex + nf * minFace > bodyBytes)
        throw std::runtime_error("PLY: the header announces " + std::to_string(nv) + " vertices and " + std::to_string(nf) + " faces, the file holds "
            + std::to_string(bodyBytes) + " bytes after it");
    std::vector<float> pos(nv * 3);
    std::vector<uint32_t> idx;
    idx.reserve(nf * 3);
    if (ascii) {
        for (size_t i = 0; i < nv; i++) {
            if (!std::getline(in, line))
                throw std::runtime_error("PLY: truncated vertex list");
            std::istringstream ls(line);
            if (!(ls >> pos[3 * i] >> pos[3 * i + 1] >> pos[3 * i + 2]))
                throw std::runtime_error("PLY: bad vertex line");
        }
        for (size_t f = 0; f < nf; f++) {
            if (!std::getline(in, line))
                throw std::runtime_error("PLY: truncated face list");
            std::istringstream ls(line);
            long n = 0;
            if (!(ls >> n) || n < 3 || n > 255) // (the binary form counts a face's vertices in one byte)
                throw std::runtime_error("PLY: a face needs 3 to 255 vertices");
            std::vector<uint32_t> poly((size_t)n);
            for (long k = 0; k < n; k++) {
                long long v = -1;
                if (!(ls >> v) || v < 0 || (uint64_t)v >= nv)
                    throw std::runtime_error("PLY: bad vertex index in a face");
                poly[(size_t)k] = (uint32_t)v;
            }
            for (long k = 1; k + 1 < n; k++) { // fan-triangulate
                idx.push_back(poly[0]);
                idx.push_back(poly[(size_t)k]);
                idx.push_back(poly[(size_t)k + 1]);
            }
        }
    } else { // all vertex properties assumed float32, faces uchar count + int32 indices
        std::vector<float> row(vertexProps);
        for (size_t i = 0; i < nv; i++) {
            if (!in.read((char*)row.data(), vertexProps * sizeof(float)))
                throw std::runtime_error("PLY: truncated vertex list");
            pos[3 * i] = row[0], pos[3 * i + 1] = row[1], pos[3 * i + 2] = row[2];
        }
        for (size_t f = 0; f < nf; f++) {
            uint8_t n = 0;
            if (!in.read((char*)&n, 1) || n < 3)
                throw std::runtime_error("PLY: truncated face list (or a face of fewer than 3 vertices)");
            std::vector<int32_t> poly(n);
            if (!in.read((char*)poly.data(), n * sizeof(int32_t)))
                throw std::runtime_error("PLY: truncated face list");
            for (int k = 0; k < n; k++)
                if (poly[k] < 0 || (uint64_t)poly[k] >= nv)
                    throw std::runtime_error("PLY: bad vertex index in a face");
            for (int k = 1; k + 1 < n; k++) {
                idx.push_back((uint32_t)poly[0]);
                idx.push_back((uint32_t)poly[k]);
                idx.push_back((uint32_t)poly[k + 1]);
            }
        }
    }
    return std::make_shared<Mesh>(pos.data(), nullptr, nullptr, nv, idx.data(), nullptr, idx.size() / 3, std::vector<Material> { material }, builder);
}

} // namespace raytracer
