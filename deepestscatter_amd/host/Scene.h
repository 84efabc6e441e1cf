// Scene.h -- host-side mirror of the reference's scene graph for the path-tracing configuration:
// SceneItem (src/Scene/SceneItem.h:5-18), Scene (Scene.h:15-31, Scene.cpp:16-62), Sun (Sun.cpp),
// VDBCloud (VDBCloud.cpp), CloudMaterial (CloudMaterial.cpp), Resources (Util/Resources.cpp),
// ARenderer / PathTracingRenderer (Scene/Cameras/*.h), Camera (Camera.cpp).
//
// Same class names, same init/update order, same error behaviour (exceptions on the host side,
// std::invalid_argument for a bad mode); the OptiX context is replaced by `Context`, which
// collects the variables the device programs used to read and owns the libcloudtrace handle.
// Everything numeric happens behind the C ABI (include/cloudtrace.h) on the GPU.
#pragma once

#include <cstdint>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/cloudtrace.h"
#include "SceneDescription.h"

namespace DeepestScatter
{
    // Stand-in for optix::Context: variable scopes + the renderer handle.
    class Context
    {
    public:
        Context() { scene.abi_version = CT_ABI_VERSION; scene.max_depth = 2000; scene.shard_count = 1; }
        ~Context() { destroy(); }
        Context(const Context&) = delete;
        Context& operator=(const Context&) = delete;

        void destroy()                                                               // GuiExecutionLoop.cpp:93-97
        {
            if (group) { ct_group_destroy(group); group = nullptr; }
            if (handle) { ct_destroy(handle); handle = nullptr; }
        }

        static void check(int rc, CtHandle h, const char* what)
        {
            if (rc != CT_OK)
            {
                throw std::runtime_error(std::string(what) + ": " + ct_last_error(h));   // optix::Exception path
            }
        }

        static void checkGroup(int rc, CtGroup g, const char* what)
        {
            if (rc != CT_OK) throw std::runtime_error(std::string(what) + ": " + ct_group_last_error(g));
        }

        // One renderer handle, or (devices.size() > 1: cloudtrace --gpus) a group of them, one per GPU, each rendering
        // its pixel tiles; the calls Camera makes go to whichever exists (the group merges the shards with RCCL).
        void create()
        {
            destroy();
            if (devices.size() > 1) checkGroup(ct_group_create(&scene, devices.data(), (uint32_t)devices.size(), &group), nullptr, "ct_group_create");
            else { if (!devices.empty()) scene.device = devices[0]; check(ct_create(&scene, &handle), nullptr, "ct_create"); }
        }
        void setCamera(const float* eye, const float* U, const float* V, const float* W)
        {
            if (group) checkGroup(ct_group_set_camera(group, eye, U, V, W), group, "ct_group_set_camera");
            else check(ct_set_camera(handle, eye, U, V, W), handle, "ct_set_camera");
        }
        void renderAccumulate(uint32_t first, uint32_t count, bool enqueue)
        {
            if (group) checkGroup(ct_group_render_accumulate(group, first, count), group, "ct_group_render_accumulate");
            else if (enqueue) check(ct_render_accumulate_async(handle, first, count), handle, "ct_render_accumulate_async");
            else check(ct_render_accumulate(handle, first, count), handle, "ct_render_accumulate");
        }
        void resetAccumulation()
        {
            if (group) checkGroup(ct_group_reset(group), group, "ct_group_reset");
            else check(ct_reset(handle), handle, "ct_reset");
        }
        void downloadMean(float* dst, size_t bytes)
        {
            if (group) checkGroup(ct_group_download(group, CT_BUF_MEAN, dst, bytes), group, "ct_group_download");
            else check(ct_download(handle, CT_BUF_MEAN, dst, bytes), handle, "ct_download");
        }
        void tonemap(float exposure, uint8_t* screen)
        {
            if (group) checkGroup(ct_group_tonemap(group, exposure, screen, nullptr), group, "ct_group_tonemap");
            else check(ct_tonemap(handle, exposure, screen, nullptr), handle, "ct_tonemap");
        }
        // Camera::render's isConverged() before every update, decided on the device (ct_set_stop_when_converged): a single
        // handle only -- a group tests its merged frame where it is read
        bool stopWhenConverged(uint32_t cadence, uint32_t minSubframes)
        {
            if (group || !handle) return false;
            check(ct_set_stop_when_converged(handle, cadence, minSubframes), handle, "ct_set_stop_when_converged");
            return true;
        }
        void wait() { if (handle) check(ct_synchronize(handle), handle, "ct_synchronize"); }
        uint32_t convergedAt(uint64_t& left)                                       // never waits
        {
            uint32_t at = 0, tested = 0;
            check(ct_converged_at(handle, &at, &tested, &left), handle, "ct_converged_at");
            return at;
        }
        bool isConverged(uint64_t& left)
        {
            int32_t converged = 0;
            if (group) checkGroup(ct_group_is_converged(group, &converged, &left), group, "ct_group_is_converged");
            else check(ct_is_converged(handle, &converged, &left), handle, "ct_is_converged");
            return converged != 0;
        }

        CtScene scene{};
        CtHandle handle = nullptr;
        CtGroup group = nullptr;
        std::vector<int32_t> devices;       // empty: device 0
        std::vector<uint8_t> density;       // filled by VDBCloud::InitVolume
        std::vector<float> mie, choppedMie; // filled by Scene::init (Mie::get*Sampler)
    };

    class SceneItem
    {
    public:
        virtual ~SceneItem() = default;
        virtual void init() = 0;
        virtual void reset() {}
        virtual void update() {}
        virtual bool isCompleted() { return true; }
    };

    class Resources
    {
    public:
        explicit Resources(std::shared_ptr<Context> context) : context(std::move(context)) {}

        // Resources::loadVolumeBuffer (Resources.cpp:68-155): returns the uint8 texture (zero border
        // included) and its size in texels.  Sources: "procedural:<N>[:<seed>]" (the synthetic
        // benchmark cloud) or a raw float grid "<file>.f32grid" = int32 nx,ny,nz + nx*ny*nz floats
        // (x fastest), which goes through the reference's quantiser, or a .vdb file -- the reference's own
        // input -- read by host/VdbReader.h (no OpenVDB needed).
        std::tuple<std::vector<uint8_t>, std::array<uint32_t, 3>> loadVolumeBuffer(const std::string& path, bool createMipmaps)
        {
            std::cout << "Loading volume... " << path << std::endl;
            (void)createMipmaps; // the pyramid is only read by the neural descriptors (out of scope)
            if (path.rfind("procedural:", 0) == 0)
            {
                uint32_t n = 0, seed = 0xC10D5EEDu;
                if (std::sscanf(path.c_str(), "procedural:%u:%u", &n, &seed) < 1) throw std::invalid_argument("bad procedural spec");
                std::vector<uint8_t> tex((size_t)n * n * n);
                if (ct_make_procedural_cloud(n, seed, tex.data()) != CT_OK) throw std::runtime_error("ct_make_procedural_cloud failed");
                return { std::move(tex), { n, n, n } };
            }
            if (path.size() > 8 && path.substr(path.size() - 8) == ".f32grid")
            {
                std::ifstream f(path, std::ios::binary);
                if (!f.good()) throw std::runtime_error("cannot open " + path);
                int32_t d[3];
                f.read(reinterpret_cast<char*>(d), sizeof d);
                std::vector<float> grid((size_t)d[0] * d[1] * d[2]);
                f.read(reinterpret_cast<char*>(grid.data()), (std::streamsize)(grid.size() * sizeof(float)));
                const uint32_t pd[3] = { (uint32_t)d[0], (uint32_t)d[1], (uint32_t)d[2] };
                std::vector<uint8_t> tex((size_t)(pd[0] + 2) * (pd[1] + 2) * (pd[2] + 2));
                if (ct_quantize_volume(grid.data(), pd, tex.data()) != CT_OK) throw std::runtime_error("ct_quantize_volume failed");
                return { std::move(tex), { pd[0] + 2, pd[1] + 2, pd[2] + 2 } };
            }
            if (path.size() > 4 && path.substr(path.size() - 4) == ".vdb")
            {
                // the reference's own input (Resources.cpp:82-143), read without OpenVDB: ct_load_vdb -> host/VdbReader.h
                uint32_t d[3] = { 0, 0, 0 };
                size_t bytes = 0;
                char err[512] = { 0 };
                if (ct_load_vdb(path.c_str(), d, nullptr, 0, &bytes, err, sizeof err) != CT_OK) throw std::runtime_error(path + ": " + err);
                std::vector<uint8_t> tex(bytes);
                if (ct_load_vdb(path.c_str(), d, tex.data(), tex.size(), &bytes, err, sizeof err) != CT_OK) throw std::runtime_error(path + ": " + err);
                std::cout << "Creating buffer of size " << d[0] << "x" << d[1] << "x" << d[2] << std::endl;   // Resources.cpp:119
                return { std::move(tex), { d[0], d[1], d[2] } };
            }
            throw std::runtime_error("unsupported volume '" + path + "' (use <file>.vdb, procedural:<N> or <file>.f32grid)");
        }

        // The Lorenz-Mie tables (data of Mie.cpp:8-8203) shipped as deepestscatter_amd/data/mie_raw.f32.
        static void loadMie(const std::string& dataDir, std::vector<float>& mie, std::vector<float>& chopped)
        {
            std::ifstream f(dataDir + "/mie_raw.f32", std::ios::binary);
            if (!f.good()) throw std::runtime_error("cannot open " + dataDir + "/mie_raw.f32");
            mie.resize(4096);
            chopped.resize(4096);
            f.read(reinterpret_cast<char*>(mie.data()), 4096 * sizeof(float));
            f.read(reinterpret_cast<char*>(chopped.data()), 4096 * sizeof(float));
        }

    private:
        std::shared_ptr<Context> context;
    };

    class Sun : public SceneItem
    {
    public:
        using Settings = DirectionalLight;
        Sun(std::shared_ptr<Settings> settings, std::shared_ptr<Context> context)
            : context(std::move(context)), direction(settings->direction), color(settings->color), intensity(settings->intensity) {}

        void init() override                                                    // Sun.cpp:13-18
        {
            for (int i = 0; i < 3; i++) { context->scene.light_direction[i] = direction[i]; context->scene.light_color[i] = color[i]; }
            context->scene.light_intensity = intensity;
            // `direction` has been normalised by installSceneSetup and by DirectionalLight's constructor
            context->scene.flags |= CT_FLAG_LIGHT_NORMALIZED;
        }

    private:
        std::shared_ptr<Context> context;
        float3 direction; Color color; float intensity;
    };

    class VDBCloud : public SceneItem
    {
    public:
        using Settings = Cloud::Model;
        VDBCloud(std::shared_ptr<Settings> settings, std::shared_ptr<Context> context, std::shared_ptr<Resources> resources)
            : settings(*settings), context(std::move(context)), resources(std::move(resources)) {}

        void init() override                                                    // VDBCloud.cpp:15-20
        {
            InitVolume();
            // InitInScatter (VDBCloud.cpp:57-86) runs inside ct_create once the Sun is known.
            setupVariables();
        }

        std::array<uint32_t, 3> getResolution() const { return dims; }
        float getVoxelSizeInMeters() const { return settings.size / (float)std::max({ dims[0], dims[1], dims[2] }); }
        float getVoxelSizeInTermsOfFreePath() const { return getVoxelSizeInMeters() / settings.meanFreePath; }

    private:
        void InitVolume()                                                        // :48-55
        {
            auto cloud = resources->loadVolumeBuffer(settings.vdbPath, static_cast<bool>(settings.mipmapsOn));
            context->density = std::move(std::get<0>(cloud));
            dims = std::get<1>(cloud);
        }

        void setupVariables()                                                    // :88-111
        {
            for (int i = 0; i < 3; i++) context->scene.dims[i] = dims[i];
            context->scene.cloud_size_m = settings.size;
            context->scene.mean_free_path_m = settings.meanFreePath;
        }

        const Settings settings;
        std::shared_ptr<Context> context;
        std::shared_ptr<Resources> resources;
        std::array<uint32_t, 3> dims{};
    };

    class CloudMaterial : public SceneItem
    {
    public:
        using Settings = std::shared_ptr<Cloud::Rendering>;
        CloudMaterial(Settings settings, std::shared_ptr<Context> context) : context(std::move(context)), renderSettings(std::move(settings))
        {
            this->context->scene.sample_step = renderSettings->sampleStep;       // CloudMaterial.cpp:14
        }

        void init() override { context->scene.mode = getRenderMode(); }

        int getRenderMode() const                                                // getRenderProgramName, :51-64
        {
            switch (renderSettings->mode)
            {
            case Cloud::Rendering::Mode::SunAndSkyAllScatter: return CT_MODE_SUN_AND_SKY_ALL_SCATTER;   // totalRadiance
            case Cloud::Rendering::Mode::SunMultipleScatter: return CT_MODE_SUN_MULTIPLE_SCATTER;       // multipleScatterSunRadiance
            case Cloud::Rendering::Mode::SunSingleScatter: return CT_MODE_SUN_SINGLE_SCATTER;           // singleScatterSunRadiance
            default: throw std::invalid_argument("Invalid Render Mode");
            }
        }

    private:
        std::shared_ptr<Context> context;
        Settings renderSettings;
    };

    class Scene
    {
    public:
        Scene(std::vector<std::shared_ptr<SceneItem>> sceneItems, std::shared_ptr<Context> context, std::string dataDir)
            : sceneItems(std::move(sceneItems)), context(std::move(context)), dataDir(std::move(dataDir)) {}

        void init()                                                              // Scene.cpp:36-46
        {
            Resources::loadMie(dataDir, context->mie, context->choppedMie);      // Mie samplers, :38-40
            for (const auto& item : sceneItems) item->init();                    // Sun, VDBCloud, CloudMaterial, Camera
        }
        void update() { for (const auto& item : sceneItems) item->update(); }    // :48-54
        bool isCompleted()                                                        // :56-62
        {
            bool done = true;
            for (const auto& item : sceneItems) done &= item->isCompleted();
            return done;
        }
        void restartProgressive() { for (const auto& item : sceneItems) item->reset(); }

    private:
        std::vector<std::shared_ptr<SceneItem>> sceneItems;
        std::shared_ptr<Context> context;
        std::string dataDir;
    };
}
